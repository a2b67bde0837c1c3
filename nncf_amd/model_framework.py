"""Drop-in for the reference's models/model_framework.py: `get_model(conf, data_helper,
model_name)` returns the same nine-key model_dict (model_framework.py:195-204) whose values
expose the Keras-Model subset the trainers and the evaluator use (`train_on_batch`,
`predict_on_batch`, `predict`), backed by the HIP kernels of libnncf_hip.so instead of a
Keras/TF graph.

Scope of this build: the id-table towers (`model_name == 'mf'`, model_framework.py:85-88, and
the user side of every model, :59-63) and the mean-pool text tower (`basic_embedding`,
:90-91,98-111; nncf_amd/mean_pool.py).  cnn_embedding / rnn_embedding raise.
All heads share ONE parameter set, as the reference's models share layer objects.
"""
import numpy as np
import torch

from . import _lib
from .engine import Engine, _ptr


def _as_ids(x):
    return np.ascontiguousarray(np.asarray(x).reshape(-1))


def _labels_positive_first(y, b):
    yy = np.asarray(y[0] if isinstance(y, (list, tuple)) else y).reshape(-1)
    return yy.size == 0 or (np.all(yy[:b] == 1) and np.all(yy[b:] != 1))


class _Head(object):
    """Common bits of the model objects: shared Engine, sync-and-return-float semantics of
    Keras' train_on_batch (the reference adds the returned host scalar to `cost`,
    train_original.py:57-58)."""

    def __init__(self, engine):
        self.engine = engine

    def _loss(self, t):
        return float(t.item())  # device -> host sync, as session.run did


class OriginalModel(_Head):
    """model_dict['model'] (model_framework.py:123,147-149): row-wise interaction, first
    batch_size_p rows positive, negative row j belongs to positive j // k."""

    def __init__(self, engine, update=True):
        super(OriginalModel, self).__init__(engine)
        self.update = update

    def train_on_batch(self, x, y=None):
        uid, cid = _as_ids(x[0]), _as_ids(x[1])
        b, k = self.engine.b, self.engine.k
        if uid.size != b * (1 + k) or cid.size != b * (1 + k):
            raise ValueError("[ERROR] model expects batch_size_p * (1 + num_negatives) = %d rows, got %d"
                             % (b * (1 + k), uid.size))
        e = self.engine
        as_written = e.optimizer == "sparse_adam" and getattr(e, "sparse_adam_duplicates", "sum") == "as_written"
        if as_written and y is None:   # every row is a lookup position of its own: the row-wise kernels
            y = np.concatenate([np.ones(b), np.full(b * k, -1 if e.cfg.loss == 0 else 0)])
        if not as_written and np.array_equal(uid[b:], np.repeat(uid[:b], k)) and (y is None or _labels_positive_first(y, b)):
            # the trainer's layout (train_original.py:49-56): per-positive kernel
            return self._loss(e.step_original(e.ids(uid[:b]), e.ids(cid), update=self.update))
        # any other row order (presample / group_sample / reverse): general row-wise kernels
        if y is None:
            raise ValueError("[ERROR] row-wise batches need their labels")
        yy = np.asarray(y[0] if isinstance(y, (list, tuple)) else y).reshape(-1)
        return self._loss(e.step_rowwise(e.ids(uid), e.ids(cid), e.ids(yy), update=self.update))

    def run_epoch_rowwise(self, uid_all, cid_all, y_all, n_steps):
        return self.engine.epoch_rowwise(uid_all, cid_all, y_all, n_steps)

    def run_epoch(self, uid_all, cid_all, n_steps, neg_all=None):
        """The trainer's inner loop (train_original.py:43-68) on device; per-step losses."""
        return self.engine.epoch_original(uid_all, cid_all, neg_all, n_steps)


class NegSharedModel(_Head):
    """model_dict['model_neg_shared'] (model_framework.py:128-129,153-155)."""

    def __init__(self, engine, update=True):
        super(NegSharedModel, self).__init__(engine)
        self.update = update

    def train_on_batch(self, x, y=None):
        e = self.engine
        return self._loss(e.step_neg_shared(e.ids(_as_ids(x[0])), e.ids(_as_ids(x[1])), update=self.update))

    def run_epoch(self, uid_all, cid_all, n_steps, neg_all=None):
        return self.engine.epoch_neg_shared(uid_all, cid_all, n_steps)

    def predict_on_batch(self, x):
        """[uid (p,), cid (q,)] -> float32 (p, q) scores (utils/objectives.py:310)."""
        e = self.engine
        return e.predict_scores(e.ids(_as_ids(x[0])), e.ids(_as_ids(x[1]))).cpu().numpy()

    def predict_on_batch_device(self, uid_dev, cid_dev, out=None):
        return self.engine.predict_scores(uid_dev, cid_dev, out=out)

    def predict_engine(self):
        """the engine whose tables predict_on_batch scores with (the fused evaluator, nncf_eval_whole, reads them directly)"""
        return self.engine


class GroupNegSharedModel(_Head):
    """model_dict['model_group_neg_shared'] (model_framework.py:131-136,158-161); needs exactly
    batch_size_p rows (row_cidx_prefx, :38-39,133-134)."""

    def train_on_batch(self, x, y=None):
        e = self.engine
        uid, cid = _as_ids(x[0]), _as_ids(x[1])
        if uid.size != e.b:
            raise ValueError("[ERROR] model_group_neg_shared needs exactly batch_size_p = %d rows" % e.b)
        return self._loss(e.step_group_neg_shared(e.ids(uid), e.ids(cid)))

    def run_epoch(self, uid_all, cid_all, n_steps, neg_all=None, n_u_out=None):
        return self.engine.epoch_group_neg_shared(uid_all, cid_all, n_steps, n_u_out=n_u_out)


class SampledNegSharedModel(_Head):
    """model_dict['model_sampled_neg_shared'] (model_framework.py:138-143,166-170): b positives
    scored row-wise plus the SAME k sampled items as negatives for every positive
    (get_sampled_neg_shared_loss, objectives.py:120-161).  That is get_original_loss with every
    positive's k negatives being the shared ones, except that this head uses the user rows
    un-normalised (U_emb_front, :64-65) -- so it runs the per-positive kernels on an engine with
    normalize_user = 0 and the negatives replicated per positive; the duplicate plan folds the b
    copies of each negative into one row update."""

    def __init__(self, engine, make_unnormalised):
        super(SampledNegSharedModel, self).__init__(engine)
        self._make = make_unnormalised
        self._e2 = None

    def _engine2(self):
        if self._e2 is None:
            self._e2 = self._make()
        return self._e2

    def train_on_batch(self, x, y=None):
        e = self._engine2()
        b, k = e.b, e.k
        uid, cid = _as_ids(x[0]), _as_ids(x[1])
        if uid.size != b + k or cid.size != b + k:
            raise ValueError("[ERROR] model_sampled_neg_shared expects batch_size_p + num_negatives = %d rows" % (b + k))
        full = np.concatenate([cid[:b], np.tile(cid[b:], b)])
        return self._loss(e.step_original(e.ids(uid[:b]), e.ids(full)))

    def run_epoch(self, uid_all, cid_all, n_steps, neg_shared_all=None):
        """neg_shared_all: int32 device [n_steps * k], the k shared negatives of every step."""
        e = self._engine2()
        b, k = e.b, e.k
        neg = neg_shared_all[:n_steps * k].view(n_steps, 1, k).expand(n_steps, b, k).reshape(-1).contiguous()
        return e.epoch_original(uid_all, cid_all, neg, n_steps)


class EmbeddingModel(_Head):
    """model_user_emb / model_item_emb (model_framework.py:173-174): the tower output for a list
    of ids.  Only the 'given' evaluation scheme uses these; rows are gathered with torch
    indexing (no arithmetic besides the configured L2 normalisation)."""

    def __init__(self, engine, side):
        super(EmbeddingModel, self).__init__(engine)
        self.side = side

    def predict_on_batch(self, x):
        import torch
        e = self.engine
        ids = e.ids(_as_ids(x[0])).long()
        tab, norm = (e.U, e.cfg.normalize_user) if self.side == "user" else (e.V, e.cfg.normalize_item)
        rows = tab[ids, :e.dim]
        if norm:
            rows = rows * torch.rsqrt(torch.clamp((rows * rows).sum(-1, keepdim=True), min=1e-12))
        return rows.cpu().numpy()


class TowerItemEmbeddingModel(_Head):
    """model_item_emb above a text tower (model_framework.py:174: Model([cid], C_emb) -- predict runs in
    the Keras inference phase): the tower's inference-mode output (BatchNorm running statistics, no
    dropout) of the REQUESTED ids, L2-normalised as configured (:109-111).  The training engine's item
    "table" is a per-batch block of tower outputs, never indexed by catalogue id."""

    def __init__(self, core):
        super(TowerItemEmbeddingModel, self).__init__(core.engine)
        self.core = core

    def predict_on_batch(self, x):
        import torch
        e, tw = self.engine, self.core.tower
        ids = e.ids(_as_ids(x[0]))
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.core.item_count):
            raise IndexError("item id out of range [0, %d)" % self.core.item_count)
        rows = torch.empty(ids.numel(), e.dim_p, dtype=torch.float32, device=e.device)
        if ids.numel():
            tw.forward(ids, train=False, out=rows)
        rows = rows[:, :e.dim]
        if e.cfg.normalize_item:
            rows = rows * torch.rsqrt(torch.clamp((rows * rows).sum(-1, keepdim=True), min=1e-12))
        return rows.cpu().numpy()


class PredPairsModel(_Head):
    """model_pred_pairs (model_framework.py:125-126,175-176): row-wise dot of GIVEN embeddings."""

    def predict(self, x, batch_size=4096):
        U, C = np.asarray(x[0], dtype=np.float32), np.asarray(x[1], dtype=np.float32)
        return np.sum(U * C, axis=1, keepdims=True)


class _LossMonitorDict(dict):
    """model_all_loss / model_neg_shared_all_loss (model_framework.py:179-193): one SGD(0.)
    model per loss name, sharing the parameters.  Built on first use."""

    LOSSES = ("skip-gram", "mse", "log-loss", "max-margin")

    def __init__(self, make):
        super(_LossMonitorDict, self).__init__()
        self._make = make

    def keys(self):
        return list(self.LOSSES)

    def __iter__(self):
        return iter(self.LOSSES)

    def __len__(self):
        return len(self.LOSSES)

    def __getitem__(self, name):
        if not dict.__contains__(self, name):
            if name not in self.LOSSES:
                raise KeyError(name)
            dict.__setitem__(self, name, self._make(name))
        return dict.__getitem__(self, name)

    def items(self):
        return [(n, self[n]) for n in self.LOSSES]

    iteritems = items


# ----------------------------------------------------------------------------------------
# heads of the content models (basic_embedding): text tower above the step kernels
# ----------------------------------------------------------------------------------------
class _TowerCore(object):
    """Shared state of the basic_embedding heads: the user table + step workspace live in an
    Engine whose item "table" is the block of tower outputs for the batch's unique items
    (model_framework.py:45-48,98-106: content model on tf.unique(cid), re-gathered by cid_x);
    item gradients are exported and sent back through the tower."""

    def __init__(self, engine, tower, item_count, engine_kw, device):
        self.engine, self.tower = engine, tower
        self.item_count = item_count
        self.gV = torch.zeros_like(engine.V)
        engine.set_item_grad_export(self.gV)   # item rows are tower outputs: export, never update
        self._eval = None
        self._eval_version = -1
        self._engine_kw, self._device = engine_kw, device

    _SCHEME = {"original": 0, "neg_shared": 1, "group_neg_shared": 2}

    def step(self, kind, uid_dev, cid_dev, update=True, loss_out=None):
        """One training step, fully asynchronous (no host sync), one C call: plans + tf.unique,
        tower forward of the unique items into the engine's item block, step kernels, tower
        backward (nncf_tower_step)."""
        e, tw = self.engine, self.tower
        out = loss_out if loss_out is not None else e._loss1
        with torch.cuda.device(e.device):
            _lib.check(e._L.nncf_tower_step(tw.handle, e._h, self._SCHEME[kind], _ptr(uid_dev), _ptr(cid_dev), _ptr(out),
                                            int(update), None, e._stream()))
        if update:
            tw.version += 1
        return out

    def epoch(self, kind, uid_all, cid_all, neg_all, n_steps, losses, n_u_out=None):
        """The trainer's inner loop in one C call (nncf_tower_epoch): every launch of every step
        is enqueued from native code, plans are made 64 steps at a time."""
        e, tw = self.engine, self.tower
        with torch.cuda.device(e.device):
            _lib.check(e._L.nncf_tower_epoch(tw.handle, e._h, self._SCHEME[kind], _ptr(uid_all), _ptr(cid_all),
                                             _ptr(neg_all), int(n_steps), _ptr(losses), _ptr(n_u_out), e._stream()))
        tw.version += 1
        return losses

    def step_rowwise(self, uid_dev, cid_dev, y_dev, update=True, loss_out=None):
        """b(1+k) independent (user, item, label) rows in any order (nncf_tower_step_rowwise)."""
        e, tw = self.engine, self.tower
        out = loss_out if loss_out is not None else e._loss1
        with torch.cuda.device(e.device):
            _lib.check(e._L.nncf_tower_step_rowwise(tw.handle, e._h, _ptr(uid_dev), _ptr(cid_dev), _ptr(y_dev), _ptr(out),
                                                    int(update), None, e._stream()))
        if update:
            tw.version += 1
        return out

    def epoch_rowwise(self, uid_all, cid_all, y_all, n_steps, losses):
        e, tw = self.engine, self.tower
        with torch.cuda.device(e.device):
            _lib.check(e._L.nncf_tower_epoch_rowwise(tw.handle, e._h, _ptr(uid_all), _ptr(cid_all), _ptr(y_all),
                                                     int(n_steps), _ptr(losses), e._stream()))
        tw.version += 1
        return losses

    def eval_engine(self):
        """Engine over (shared U, inference-mode embeddings of ALL items), refreshed when the
        tower changed (predict_on_batch runs in the Keras inference phase: BN running stats,
        no dropout; objectives.py:296-321)."""
        if self._eval is None:
            kw = dict(self._engine_kw)
            kw["item_count"] = self.item_count
            Vall = torch.zeros(self.item_count, self.engine.dim_p, dtype=torch.float32, device=self.engine.device)
            self._eval = Engine(device=self._device, tables=(self.engine.U, Vall), bias_state=self.engine.bias, **kw)
        if self._eval_version != self.tower.version:
            ids = torch.arange(self.item_count, dtype=torch.int32, device=self.engine.device)
            for i0 in range(0, self.item_count, 4096):
                self.tower.forward(ids[i0:i0 + 4096], train=False, out=self._eval.V[i0:i0 + 4096])
            self._eval_version = self.tower.version
        return self._eval


class TowerHead(_Head):
    def __init__(self, core, kind, update=True):
        super(TowerHead, self).__init__(core.engine)
        self.core, self.kind, self.update = core, kind, update

    def train_on_batch(self, x, y=None):
        e = self.engine
        uid, cid = _as_ids(x[0]), _as_ids(x[1])
        if self.kind == "original":
            b, k = e.b, e.k
            if uid.size != b * (1 + k) or cid.size != b * (1 + k):
                raise ValueError("[ERROR] model expects batch_size_p * (1 + num_negatives) = %d rows, got %d"
                                 % (b * (1 + k), uid.size))
            if not (np.array_equal(uid[b:], np.repeat(uid[:b], k)) and (y is None or _labels_positive_first(y, b))):
                # any other row order (presample / group_sample / reverse): general row-wise kernels
                if y is None:
                    raise ValueError("[ERROR] row-wise batches need their labels")
                yy = np.asarray(y[0] if isinstance(y, (list, tuple)) else y).reshape(-1)
                return self._loss(self.core.step_rowwise(e.ids(uid), e.ids(cid), e.ids(yy), update=self.update))
            uid = uid[:b]
        elif uid.size != e.b:
            raise ValueError("[ERROR] %s needs exactly batch_size_p = %d rows" % (self.kind, e.b))
        return self._loss(self.core.step(self.kind, e.ids(uid), e.ids(cid), update=self.update))

    def run_epoch(self, uid_all, cid_all, n_steps, neg_all=None, n_u_out=None):
        """The trainer's inner loop: one native call per epoch (loss monitors, which never update,
        go batch by batch)."""
        e = self.engine
        b, k = e.b, e.k
        losses = torch.empty(max(n_steps, 1), dtype=torch.float32, device=e.device)
        if n_steps == 0:
            return losses[:0]
        if self.update:
            return self.core.epoch(self.kind, uid_all, cid_all, neg_all, n_steps, losses, n_u_out)[:n_steps]
        for t in range(n_steps):
            cid = cid_all[t * b:(t + 1) * b]
            if self.kind == "original":
                cid = torch.cat([cid, neg_all[t * b * k:(t + 1) * b * k]])
            self.core.step(self.kind, uid_all[t * b:(t + 1) * b], cid, update=self.update, loss_out=losses[t:t + 1])
        return losses[:n_steps]

    def run_epoch_rowwise(self, uid_all, cid_all, y_all, n_steps):
        """presample / group_sample / reverse above a text tower: n_steps batches of b(1+k) rows."""
        e = self.engine
        losses = torch.empty(max(n_steps, 1), dtype=torch.float32, device=e.device)
        if n_steps == 0:
            return losses[:0]
        if self.update:
            return self.core.epoch_rowwise(uid_all, cid_all, y_all, n_steps, losses)[:n_steps]
        n = e.b * (1 + e.k)
        for t in range(n_steps):
            sl = slice(t * n, (t + 1) * n)
            self.core.step_rowwise(uid_all[sl], cid_all[sl], y_all[sl], update=False, loss_out=losses[t:t + 1])
        return losses[:n_steps]

    def predict_on_batch(self, x):
        ev = self.core.eval_engine()
        return ev.predict_scores(ev.ids(_as_ids(x[0])), ev.ids(_as_ids(x[1]))).cpu().numpy()

    def predict_on_batch_device(self, uid_dev, cid_dev, out=None):
        return self.core.eval_engine().predict_scores(uid_dev, cid_dev, out=out)

    def predict_engine(self):
        return self.core.eval_engine()   # shared user table + the tower's inference rows of ALL items (refreshed when it changed)


class TowerSampledNegSharedHead(_Head):
    """model_dict['model_sampled_neg_shared'] above a text tower: as SampledNegSharedModel -- the
    per-positive step with the k shared negatives replicated per positive, on a core whose engine
    leaves the user rows un-normalised (U_emb_front, model_framework.py:64-65)."""

    def __init__(self, core, make_core):
        super(TowerSampledNegSharedHead, self).__init__(core.engine)
        self._make, self._core = make_core, None

    @property
    def core(self):
        if self._core is None:
            self._core = self._make()
        return self._core

    def train_on_batch(self, x, y=None):
        core = self.core
        e = core.engine
        b, k = e.b, e.k
        uid, cid = _as_ids(x[0]), _as_ids(x[1])
        if uid.size != b + k or cid.size != b + k:
            raise ValueError("[ERROR] model_sampled_neg_shared expects batch_size_p + num_negatives = %d rows" % (b + k))
        full = np.concatenate([cid[:b], np.tile(cid[b:], b)])
        return self._loss(core.step("original", e.ids(uid[:b]), e.ids(full)))

    def run_epoch(self, uid_all, cid_all, n_steps, neg_shared_all=None):
        core = self.core
        e = core.engine
        b, k = e.b, e.k
        losses = torch.empty(max(n_steps, 1), dtype=torch.float32, device=e.device)
        if n_steps == 0:
            return losses[:0]
        neg = neg_shared_all[:n_steps * k].view(n_steps, 1, k).expand(n_steps, b, k).reshape(-1).contiguous()
        return core.epoch("original", uid_all, cid_all, neg, n_steps, losses)[:n_steps]


def engine_kwargs(conf, data_spec, model_name):
    normalize = bool(conf.emb_normalization)
    return dict(
        user_count=data_spec.user_count, item_count=data_spec.item_count, dim=conf.user_dim,
        batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
        loss_gamma=getattr(conf, "loss_gamma", 1), neg_loss_weight=getattr(conf, "neg_loss_weight", 1),
        u_reg=conf.u_reg, learn_rate=abs(conf.learn_rate),
        normalize_user=normalize,                              # model_framework.py:62-63
        normalize_item=normalize and model_name != "mf",        # :109-111 (content branch only)
        interaction_bias=getattr(conf, "interaction_bias", None),   # :119-121
        bias_item_count=data_spec.item_count)


def _conv_spec(conf):
    """The convolutional front of cnn_embedding as this build runs it (cnn_model.py:61-111): one convolution layer --
    a single odd filter length (the 'best' preset, filter_lengths [3]) or a bank of up to four odd lengths whose
    outputs are concatenated (the 'default' preset, filter_lengths [[1, 3, 5]]; :63-81) -- BatchNorm, relu,
    whole-length average or max pooling (:100-109)."""
    fls, nfs = list(conf.filter_lengths), list(conf.num_filters)
    bank = None
    if len(fls) == 1:
        bank = [int(v) for v in fls[0]] if isinstance(fls[0], (list, tuple)) else [int(fls[0])]
    ok = (bank is not None and 1 <= len(bank) <= 4 and all(v % 2 == 1 and 1 <= v <= 7 for v in bank) and len(nfs) == 1 and
          list(conf.poolings) in (["average"], ["max"]) and list(conf.pool_lengths)[0] <= 0 and conf.conv_activation == "relu" and
          conf.conv_batch_normalization and not getattr(conf, "no_BN", False) and
          not getattr(conf, "contextual_spatial_gated_input", None) and
          not getattr(conf, "contextual_temporal_gated_input", None))
    if not ok:
        raise NotImplementedError(
            "cnn_embedding is built for ONE convolution layer (filter_lengths=[3], the 'best' preset, or a bank of up "
            "to four odd lengths <= 7 such as [[1, 3, 5]], the 'default' preset), conv BatchNorm + relu, "
            "poolings=['average'] or ['max'] over the whole length, no gating; got filter_lengths=%r num_filters=%r poolings=%r "
            "pool_lengths=%r" % (fls, nfs, conf.poolings, conf.pool_lengths))
    return dict(filters=int(nfs[0]), filter_len=bank[0], filter_lens=bank, dropout=float(conf.conv_dropout_rate),
                pooling=str(list(conf.poolings)[0]))


def _get_model_basic_embedding(conf, data_helper, device, seed, conv=None):
    from .mean_pool import MeanPoolTower
    data_spec = data_helper.data_spec
    if getattr(conf, "pretrain", None):
        raise NotImplementedError("the text towers are built for pretrain=None (what the demo scripts run: the pretrained "
                                  "sentence vectors are not in the repo)")
    pooling = getattr(conf, "pooling", "average")
    assert conv is not None or pooling in ("average", "max"), "[ERROR] unknown pooling: %s" % pooling   # mean_pool.py:76
    idt = conf.item_dense_transform
    # mean_pool.py:78-101 / cnn_model.py:112-130: Dense(dense_hidden_dim) -> BatchNorm -> Activation -> Dropout.  The tower's
    # output is the item row: dense_hidden_dim must be item_dim; a tower WITHOUT the transform (or with conf.no_BN) is not built.
    # No transform at all (item_dense_transform falsy): the pooled block is the item row -- its width must be item_dim.
    # conf.no_BN (mean_pool.py:88-93 `assert conf.no_BN == True`: any other value, or no attribute, keeps the BatchNormalization)
    no_bn = getattr(conf, "no_BN", None) == True   # noqa: E712  (the reference's own comparison)
    if idt and idt["dense_hidden_dim"] != conf.item_dim:
        raise NotImplementedError("item_dense_transform must be Dense(item_dim), or absent (got %r)" % (idt,))
    if no_bn and conv is not None:
        raise NotImplementedError("conf.no_BN is built for basic_embedding (cnn_model.py:93 would drop the convolution's "
                                  "BatchNormalization too: not built)")
    if not idt:
        pooled = int(list(conf.num_filters)[0]) * len(conv["filter_lens"]) if conv is not None else conf.word_dim
        if pooled != conf.item_dim:
            raise ValueError("without item_dense_transform the pooled width (%d) must equal item_dim (%d)" % (pooled, conf.item_dim))
    kw = engine_kwargs(conf, data_spec, "basic_embedding")
    cap = conf.batch_size_p * (1 + conf.num_negatives)          # most unique items a batch can hold
    kw["item_count"] = cap
    engine = Engine(device=device, seed=seed, **kw)
    tower = MeanPoolTower(data_helper.data["C"], data_spec.word_count, conf.word_dim, conf.item_dim,
                          abs(conf.learn_rate), engine.device, seed=seed,
                          dropout_rate=getattr(conf, "word_emb_dropout_rate", 0.), max_rows=cap, conv=conv, dense=bool(idt))
    if conv is None and pooling == "max":
        tower.set_pooling("max")                                        # mean_pool.py:70-72
    if conv is not None and conv.get("pooling") == "max":
        tower.set_pooling("max")                                        # cnn_model.py:107-109
    if getattr(conf, "use_content_id", False):                          # mean_pool.py:102-108
        tower.set_content_id(getattr(conf, "v_reg", 0.0), seed=seed)
    if idt and (idt["dense_hidden_actv"] != "relu" or float(idt["dense_hidden_dropout"]) != 0.):   # mean_pool.py:96-97
        tower.set_dense_tail(idt["dense_hidden_actv"], float(idt["dense_hidden_dropout"]))
    if idt and no_bn:                                                   # mean_pool.py:88-93
        tower.set_no_bn(True)
    if float(getattr(conf, "c_reg", 0) or 0) != 0.0:                    # mean_pool.py:56-59, cnn_model.py:45-48
        tower.set_c_reg(conf.c_reg)
    core = _TowerCore(engine, tower, data_spec.item_count, kw, device)
    opt = getattr(conf, "optimizer", "sgd")
    if opt != "sgd":
        if getattr(conf, "sparse_adam_duplicates", "sum") != "sum":
            raise NotImplementedError("sparse_adam_duplicates = 'as_written' is built for the id tables (model_choice = mf); "
                                      "text-tower word tables keep the summed form")
        engine.set_optimizer(opt)   # after the export buffer is set: no item-table moments
        tower.set_optimizer(opt)

    def monitor(kind):
        def make(loss_name):
            kw2 = dict(kw)
            kw2["loss"] = loss_name
            kw2["learn_rate"] = 0.0
            core2 = _TowerCore(Engine(device=device, tables=(engine.U, engine.V), bias_state=engine.bias, **kw2), tower, data_spec.item_count,
                               kw2, device)
            return TowerHead(core2, kind, update=False)
        return make

    def make_unnormalised_core():
        if not kw["normalize_user"]:
            return core
        kw2 = dict(kw, normalize_user=False)
        e2 = Engine(device=device, tables=(engine.U, engine.V), bias_state=engine.bias, **kw2)
        core2 = _TowerCore(e2, tower, data_spec.item_count, kw2, device)
        if opt != "sgd":
            e2.set_optimizer(opt, state=engine.opt_state)   # one optimiser for all heads (model_framework.py:24)
        return core2

    return {
        "model": TowerHead(core, "original"),
        "model_neg_shared": TowerHead(core, "neg_shared"),
        "model_group_neg_shared": TowerHead(core, "group_neg_shared"),
        "model_sampled_neg_shared": TowerSampledNegSharedHead(core, make_unnormalised_core),
        "model_user_emb": EmbeddingModel(engine, "user"),
        "model_item_emb": TowerItemEmbeddingModel(core),
        "model_pred_pairs": PredPairsModel(engine),
        "model_all_loss": _LossMonitorDict(monitor("original")),
        "model_neg_shared_all_loss": _LossMonitorDict(monitor("neg_shared")),
    }


def get_model(conf, data_helper, model_name, device=0, seed=0):
    """get_model (model_framework.py:18-206)."""
    data_spec = data_helper.data_spec
    assert conf.user_dim == conf.item_dim, "user_dim and item_dim must match for the dot interaction"
    if model_name == "basic_embedding":
        return _get_model_basic_embedding(conf, data_helper, device, seed)
    if model_name == "cnn_embedding":   # model_framework.py:92-93
        return _get_model_basic_embedding(conf, data_helper, device, seed, conv=_conv_spec(conf))
    if model_name != "mf":
        assert model_name in ("rnn_embedding", "pretrained"), \
            "[ERROR] Model name {} unknown".format(model_name)  # model_framework.py:97
        raise NotImplementedError(
            "model_choice=%s: the RNN item tower (modules/content/rnn_model.py) is not built; "
            "use cnn_embedding, basic_embedding (text towers) or mf (id-table towers)" % model_name)
    kw = engine_kwargs(conf, data_spec, model_name)
    engine = Engine(device=device, seed=seed, **kw)
    opt = getattr(conf, "optimizer", "sgd")
    dup = getattr(conf, "sparse_adam_duplicates", "sum")
    if opt != "sgd":
        engine.set_optimizer(opt)
        engine.set_sparse_adam_duplicates(dup)

    def make_unnormalised():
        e2 = Engine(device=device, share=engine, **dict(kw, normalize_user=False))
        if opt != "sgd":
            e2.set_optimizer(opt, state=engine.opt_state)   # one optimiser for all heads (model_framework.py:24)
            e2.set_sparse_adam_duplicates(dup)
        return e2

    def monitor(head_cls):
        def make(loss_name):
            kw2 = dict(kw)
            kw2["loss"] = loss_name
            kw2["learn_rate"] = 0.0
            return head_cls(Engine(device=device, share=engine, **kw2), update=False)
        return make

    model_dict = {
        "model": OriginalModel(engine),
        "model_neg_shared": NegSharedModel(engine),
        "model_group_neg_shared": GroupNegSharedModel(engine),
        "model_sampled_neg_shared": SampledNegSharedModel(engine, make_unnormalised),
        "model_user_emb": EmbeddingModel(engine, "user"),
        "model_item_emb": EmbeddingModel(engine, "item"),
        "model_pred_pairs": PredPairsModel(engine),
        "model_all_loss": _LossMonitorDict(monitor(OriginalModel)),
        "model_neg_shared_all_loss": _LossMonitorDict(monitor(NegSharedModel)),
    }
    return model_dict
