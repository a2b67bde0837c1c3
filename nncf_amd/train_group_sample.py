"""models/train_group_sample.py:8-108 with group_shuffling_trick = True (:66-83): links
stratified by item (or user), and per batch k negatives per positive on the OTHER side
(negative users for item groups, negative items for user groups).  The reference's by-user
branch writes the sampled items into the wrong array (train_batch instead of train_batch_n,
:82); the intended assignment is implemented."""
import time

import numpy as np

from .data_utils import GroupSampler, group_shuffle_train
from .train_rowwise import RowwiseTrainer


class Trainer(RowwiseTrainer):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        assert conf.loss not in ["log-loss", "max-margin"], "[ERROR] group_sample does not support pairwise losses"
        self.group_shuffling_trick = getattr(conf, "group_shuffling_trick", False)
        if not self.group_shuffling_trick:   # train_group_sample.py:30-35
            self.group_sampler = GroupSampler(data_helper.data["train"], group_by="item", chop=conf.chop_size,
                                              neg_dist=conf.neg_dist, neg_sign=self.neg_sign,
                                              device=self.model_train.engine.device.index)
            return
        self.sample_batch = data_helper.sampler_dict["sample_batch"]
        self.sample_batch_u = data_helper.sampler_dict["sample_batch_u"]
        _num_in_train = np.max(data_helper.data["train"], axis=0) + 1
        self._iidx = {"user": np.arange(_num_in_train[0]), "item": np.arange(_num_in_train[1])}
        st = conf.shuffle_st
        assert st.startswith("by_user") or st.startswith("by_item")
        self.by = "user" if st.startswith("by_user") else "item"
        print("[INFO] sampling group based on {}".format(self.by))

    def ragged_tail(self, rows, n):
        return len(self.data_helper.data["train"]) % self.conf.batch_size_p != 0

    def build_epoch(self, epoch):
        conf = self.conf
        b, k = conf.batch_size_p, conf.num_negatives
        if not self.group_shuffling_trick:   # :86-88: every batch straight from the GroupSampler
            n_steps = len(self.data_helper.data["train"]) // b if epoch > 0 else 0
            t0 = time.time()
            if getattr(conf, "device_shuffle", False):
                # the epoch's batches drawn on the device (nncf_group_sample_with_negs): no host loop, no upload
                import torch
                self._epoch_seed = getattr(self, "_epoch_seed", 0) + 1
                uid, cid, y = self.group_sampler.sample_with_negs_device(b, k, n_steps, seed=self._epoch_seed)
                torch.cuda.synchronize(uid.device)
                return (uid, cid, y), time.time() - t0
            rows = [self.group_sampler.sample_with_negs(b, k) for _ in range(n_steps)]
            rows = np.concatenate(rows, 0) if rows else np.zeros((0, 3), dtype=np.int64)
            return rows, time.time() - t0
        train = group_shuffle_train(self.data_helper.data["train"], by=self.by, chop=conf.chop_size,
                                    iidx=self._iidx[self.by])
        self.data_helper.data["train"] = train
        n_steps = len(train) // b if epoch > 0 else 0   # nothing is drawn at epoch 0 (no training there)
        t0 = time.time()
        pos = train[:n_steps * b].reshape(n_steps, b, 3)
        neg = pos.repeat(k, axis=1)                                  # (steps, b*k, 3): row j <-> positive j // k
        draws = (self.sample_batch_u if self.by == "item" else self.sample_batch)(n_steps * b * k)
        neg[:, :, 0 if self.by == "item" else 1] = draws.reshape(n_steps, b * k)   # per-step draws, same stream
        neg[:, :, 2] = self.neg_sign
        rows = np.concatenate([pos, neg], axis=1).reshape(-1, 3)
        return rows, time.time() - t0
