"""Option surface of the hot path: mirrors the reference's Conf attribute bags
(configs/basic_embedding_conf.py:10-86 for basic_embedding, configs/pretrained_conf.py:10-68
for mf) and get_conf (:170-179).  Only fields consumed on the sampling + scoring path are kept;
text-tower presets (pretrain files, gating) are out of scope.

Optimizer: `conf.optimizer` is a string here (the reference stores a Keras / TF object):
  'adam'         Keras Adam(learn_rate), what the reference compiles for learn_rate > 0
                 (basic_embedding_conf.py:59-60) -- the default, as in the reference: dense, every row
                 of every table moves each step;
  'sparse_adam'  utils/optimizer.py AdamOptimizer(-learn_rate), what the reference switches to for
                 learn_rate < 0 (:61-66); chosen automatically for a negative learn_rate
                 (conf.sparse_adam_duplicates = 'sum' | 'as_written' picks its duplicate-id semantics);
  'sgd'          sparse SGD scatter update (BASELINE north-star) -- opt-in: {'optimizer': 'sgd'} in
                 param_dict; bench.py's headline sets it explicitly.
"""


class Conf(object):
    """configs/basic_embedding_conf.py:10-56 (defaults), param_dict override semantics :48-50."""

    model_choice = "basic_embedding"

    def __init__(self, data_name, param_dict=None):
        self.data_name = data_name

        # loss function related (basic_embedding_conf.py:15-27)
        self.max_epoch = 40
        self.batch_size_p = 512
        self.num_negatives = 10
        self.loss = "skip-gram"
        # NB: as in the reference these three are computed from the DEFAULT loss, before
        # param_dict is applied (:22-24) -- scripts always pass them explicitly.
        self.learn_rate = 0.001 if self.loss == "mse" else 0.01
        self.loss_gamma = 0.1 if self.loss == "max-margin" else 10
        self.neg_loss_weight = 8 if self.loss == "mse" else 128
        self.neg_dist = "unigram"
        self.neg_sampling_power = 1
        self.emb_normalization = None  # set below

        # train scheme related (:29-32)
        self.shuffle_st = "by_item_chop"
        self.chop_size = 2
        self.group_shuffling_trick = True

        # embedding related (:34-40)
        self.user_dim = self.item_dim = 50
        self.word_dim = 50
        self.u_reg = 1e-6
        self.c_reg = 0
        self.word_emb_dropout_rate = 0.
        self.pooling = "average"

        # uninterested (:42-46)
        self.eval_topk = 50
        self.interaction_bias = None
        self.use_content_id = False
        self.v_reg = 0

        if param_dict is not None:
            self.__dict__.update(param_dict)
        self._post_init()

    def _post_init(self):
        if self.emb_normalization is None:  # :52-56
            self.emb_normalization = True if self.loss == "max-margin" or self.loss == "log-loss" else False
        opt = self.__dict__.get("optimizer")
        if opt is None:
            # :59-66: Adam(learn_rate) for learn_rate > 0, utils/optimizer.py AdamOptimizer(-learn_rate) for
            # learn_rate < 0.  The north-star's sparse SGD scatter update is opt-in: {'optimizer': 'sgd'}
            opt = "adam" if self.learn_rate > 0 else "sparse_adam"
        assert opt in ("sgd", "adam", "sparse_adam"), "[ERROR] optimizer %r unknown" % (opt,)
        self.optimizer = opt
        # duplicate ids of a batch under 'sparse_adam': 'sum' (TF sums the IndexedSlices first) | 'as_written'
        # (utils/optimizer.py:108-134 taken literally: last-writer-wins moments, one scatter_sub per occurrence)
        dup = self.__dict__.get("sparse_adam_duplicates") or "sum"
        assert dup in ("sum", "as_written"), "[ERROR] sparse_adam_duplicates %r unknown" % (dup,)
        self.sparse_adam_duplicates = dup
        self.pretrain = None
        self.item_dense_transform = {"dense_hidden_dim": self.user_dim, "dense_hidden_dropout": 0.,
                                     "dense_hidden_actv": "relu"}  # :83-86


class CnnConf(Conf):
    """configs/cnn_embedding_conf.py:10-93 (defaults) for model_choice = cnn_embedding."""

    model_choice = "cnn_embedding"

    def __init__(self, data_name, param_dict=None):
        Conf.__init__(self, data_name, param_dict=None)
        self.max_epoch = 40
        self.u_reg = 1e-5
        # content model related (cnn_embedding_conf.py:41-48)
        self.num_filters = [50]
        self.filter_lengths = [[1, 3, 5]]
        self.poolings = ["average"]
        self.pool_lengths = [-1]
        self.conv_dropout_rate = 0.3
        self.conv_activation = "relu"
        self.conv_batch_normalization = True
        if param_dict is not None:
            self.__dict__.update(param_dict)
        self.emb_normalization = self.__dict__.get("emb_normalization") if param_dict and "emb_normalization" in param_dict else None
        self._post_init()

    def _post_init(self):
        Conf._post_init(self)
        self.contextual_spatial_gated_input = None
        self.contextual_temporal_gated_input = None


class MfConf(Conf):
    """configs/pretrained_conf.py:10-41 -- the conf main.py uses for model_choice == 'mf' (:48-49)."""

    model_choice = "mf"

    def __init__(self, data_name, param_dict=None):
        self.data_name = data_name
        self.max_epoch = 20
        self.num_negatives = 5
        self.batch_size_p = 64
        self.eval_topk = 50
        self.loss = "skip-gram"
        self.neg_loss_weight = 1
        self.loss_gamma = 10          # not set by pretrained_conf; objectives.py:26-32 falls back to 1 with a warning
        self.use_content_id = False
        self.v_reg = 0
        self.interaction_multiplier = False
        self.interaction_bias = "item"
        self.emb_normalization = None
        self.learn_rate = 0.01
        self.user_dim = self.item_dim = 50
        self.word_dim = 50
        self.u_reg = 1e-5
        self.c_reg = 0
        self.neg_dist = "uniform"
        self.neg_sampling_power = 1
        self.chop_size = 1
        self.shuffle_st = "by_item"
        self.group_shuffling_trick = True
        self.word_emb_dropout_rate = 0.
        self.pooling = "average"
        self.evaluation_mode = False
        if param_dict is not None:
            self.__dict__.update(param_dict)
        self._post_init()


def _get_conf_base(param_dict):
    """configs/conf_utils.py:1-10"""
    param_dict = param_dict or {}
    return param_dict.get("conf_seed"), param_dict.get("conf_var")


def get_conf(data_name, conf_choice, param_dict=None, model_choice="basic_embedding"):
    """get_conf (basic_embedding_conf.py:170-179 / pretrained_conf.py:167-178).

    'default': Conf(param_dict).  'best': the citeulike preset (max_epoch 30, u_reg 1e-6,
    num_negatives 10; word dropout 0.5 for the text tower) and -- when param_dict contains
    'reset_after_getconf' -- param_dict applied once more on top (:137-141)."""
    cls = MfConf if model_choice == "mf" else CnnConf if model_choice == "cnn_embedding" else Conf
    conf_seed, conf_var = _get_conf_base(param_dict)
    if conf_choice == "default":
        conf = cls(data_name, param_dict=param_dict)
    elif conf_choice == "best":
        conf = cls(data_name, param_dict=param_dict)
        conf.c_reg = 0
        conf.num_negatives = 10
        conf.max_epoch = 30
        conf.u_reg = 1e-6
        if cls is MfConf:
            conf.interaction_bias = None
        elif cls is CnnConf:   # cnn_embedding_conf.py:107-130 (citeulike_title_and_abstract)
            conf.num_filters = [50]
            conf.word_emb_dropout_rate = 0.5
            conf.filter_lengths = [3]
            conf.conv_dropout_rate = 0.1
        else:
            conf.word_emb_dropout_rate = 0.5
        if param_dict is not None and "reset_after_getconf" in param_dict:
            conf.__dict__.update(param_dict)
    else:
        assert False, "[Error!] conf_choice {} unknonw".format(conf_choice)
    if conf_seed:
        conf.conf_seed = conf_seed
    if conf_var:
        conf.conf_var = conf_var
    conf.model_choice = cls.model_choice
    return conf
