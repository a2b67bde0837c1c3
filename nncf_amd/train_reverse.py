"""models/train_reverse.py:8-83 -- given a link, sample negative USERS (:59-63)."""
import time

import numpy as np

from .train_base import shuffle_rows
from .train_rowwise import RowwiseTrainer


class Trainer(RowwiseTrainer):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        assert conf.loss not in ["log-loss", "max-margin"], "[ERROR] revrese does not support pairwise losses"
        self.sample_batch_u = data_helper.sampler_dict["sample_batch_u"]

    def ragged_tail(self, rows, n):
        return len(self.data_helper.data["train"]) % self.conf.batch_size_p != 0

    def build_epoch(self, epoch):
        conf = self.conf
        b, k = conf.batch_size_p, conf.num_negatives
        train = self.data_helper.data["train"]
        shuffle_rows(train)
        n_steps = len(train) // b if epoch > 0 else 0   # nothing is drawn at epoch 0 (no training there)
        t0 = time.time()
        pos = train[:n_steps * b].reshape(n_steps, b, 3)
        neg = pos.repeat(k, axis=1)
        neg[:, :, 0] = self.sample_batch_u(n_steps * b * k).reshape(n_steps, b * k)
        neg[:, :, 2] = self.neg_sign
        return np.concatenate([pos, neg], axis=1).reshape(-1, 3), time.time() - t0
