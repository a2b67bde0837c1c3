"""Shared machinery of the trainers that feed model.train_on_batch arbitrary (user, item, label)
rows: presample (models/train_presample.py), group_sample (train_group_sample.py) and reverse
(train_reverse.py).  Each epoch's batches are assembled on the host exactly as the reference
does (numpy, same RNG calls, same sampler stream), uploaded once, and the inner loop runs on
the device in nncf_epoch_rowwise (general row-wise kernels, csrc/step_rowwise.hip)."""
import time

import numpy as np
import torch

from .train_base import TrainerBase


class RowwiseTrainer(TrainerBase):
    pairwise_ok = False

    def __init__(self, model_dict, conf, data_helper):
        super(RowwiseTrainer, self).__init__(model_dict, conf, data_helper)
        self.model_train = model_dict["model"]
        self.neg_sign = -1 if conf.loss == "skip-gram" else 0

    def ragged_tail(self, rows, n):
        """does the reference's loop count one broken-off iteration this epoch?  presample walks the
        pre-sampled array itself; group_sample / reverse walk the link list (overridden there)"""
        return len(rows) % n != 0

    def build_epoch(self, epoch):
        """-> (rows (n_rows, 3) int64, per-epoch host cost excluded from the timing window)."""
        raise NotImplementedError

    def train(self, eval_scheme=None, use_async_eval=True):
        conf = self.conf
        n = conf.batch_size_p * (1 + conf.num_negatives)
        engine = self.model_train.engine
        self.train_time = []
        for epoch in range(conf.max_epoch + 1):
            rows, in_window = self.build_epoch(epoch)
            cost, it = 0, 0
            if isinstance(rows, tuple):      # (uid, cid, y) int32 device tensors: the epoch was assembled on the device
                uid, cid, y = rows
                n_steps = uid.numel() // n if epoch > 0 else 0
                rows = np.zeros((uid.numel(), 0))
            else:
                n_steps = len(rows) // n if epoch > 0 else 0   # ragged tail dropped
                m = n_steps * n
                to = lambda col: torch.as_tensor(np.ascontiguousarray(rows[:m, col]).astype(np.int32)).to(engine.device)  # noqa: E731
                uid, cid, y = to(0), to(1), to(2)
            torch.cuda.synchronize(engine.device)
            t_start = time.time()
            if n_steps:
                losses = self.model_train.run_epoch_rowwise(uid, cid, y, n_steps)
                cost = float(losses.double().sum().item())
                it = n_steps + (1 if n_steps and self.ragged_tail(rows, n) else 0)   # train_presample.py:88-93
            if epoch > 0:
                # sampling done per step inside the reference's loop is part of its timing window
                self.train_time.append(time.time() - t_start + in_window)
            self._epoch_report(epoch, it, cost, eval_scheme)
        self._finish()
