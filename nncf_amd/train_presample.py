"""models/train_presample.py:8-114 -- negatives pre-sampled for the whole epoch, then shuffled /
grouped according to conf.shuffle_st (:44-85)."""
import numpy as np

from .data_utils import group_shuffle_train
from .train_base import shuffle_rows
from .train_rowwise import RowwiseTrainer


class Trainer(RowwiseTrainer):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        self.sample_batch = data_helper.sampler_dict["sample_batch"]
        self.sample_batch_u = data_helper.sampler_dict["sample_batch_u"]
        _num_in_train = np.max(data_helper.data["train"], axis=0) + 1
        self._iidx = {"user": np.arange(_num_in_train[0]), "item": np.arange(_num_in_train[1])}
        assert conf.loss not in ["log-loss", "max-margin"], \
            "[ERROR] shuffle_st %s does not support pairwise losses" % conf.shuffle_st  # :21-25 (all of them here)

    def build_epoch(self, epoch):
        conf = self.conf
        k, chop, neg_sign = conf.num_negatives, conf.chop_size, self.neg_sign
        train_p = self.data_helper.data["train"]
        st = conf.shuffle_st
        if st == "original":  # :44-50
            shuffle_rows(train_p)
            train = train_p.repeat(1 + k, axis=0)
            train[:, 1] = self.sample_batch(train.shape[0])
            train[:, 2] = neg_sign
            train[::1 + k] = train_p
        elif st == "reverse":  # :51-57
            shuffle_rows(train_p)
            train = train_p.repeat(1 + k, axis=0)
            train[:, 0] = self.sample_batch_u(train.shape[0])
            train[:, 2] = neg_sign
            train[::1 + k] = train_p
        else:  # :58-85
            train_n = train_p.repeat(k, axis=0)
            train_n[:, 1] = self.sample_batch(train_n.shape[0])
            train_n[:, 2] = neg_sign
            train = np.vstack((train_p, train_n))
            if st == "random":
                shuffle_rows(train)
            elif st == "by_user":
                train = group_shuffle_train(train, by="user", iidx=self._iidx["user"])
            elif st == "by_item":
                train = group_shuffle_train(train, by="item", iidx=self._iidx["item"])
            elif st.startswith("by_user_chop"):
                train = group_shuffle_train(train, by="user", chop=chop, iidx=self._iidx["user"])
            elif st.startswith("by_item_chop"):
                train = group_shuffle_train(train, by="item", chop=chop, iidx=self._iidx["item"])
            elif st.startswith("by_useritem_chop"):
                by = "user" if epoch % 2 == 0 else "item"
                train = group_shuffle_train(train, by=by, chop=chop, iidx=self._iidx[by])
            else:
                assert False, "ERROR: unknown shuffle strategy {}".format(st)
        return train, 0.0  # pre-sampling sits outside the reference's timing window (:87)
