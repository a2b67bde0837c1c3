"""models/train_group_neg_shared.py:7-76 -- group (stratified) negative sharing.
Per epoch the links are re-ordered by group_shuffle_train (item strata, chop) on the host
(outside the timing window, as the reference: :42-46); the inner loop (:53-68) runs on the
device inside nncf_epoch_group_neg_shared, which also de-duplicates each batch's items
(tf.unique, model_framework.py:45-48)."""
import time

import numpy as np
import torch

from .data_utils import GroupSampler, group_shuffle_train, group_shuffle_train_device
from .train_base import TrainerBase


class Trainer(TrainerBase):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        self.model_train = model_dict["model_group_neg_shared"]
        if conf.neg_dist != "unigram":
            print("[WARNING] Only unigram neg_dist is currently supported "
                  "for group_neg_shared training. Set neg_dist = unigram.")
        self.group_shuffling_trick = getattr(conf, "group_shuffling_trick", False)
        if self.group_shuffling_trick:
            _num_in_train = np.max(data_helper.data["train"], axis=0) + 1
            self._iidx = {"user": np.arange(_num_in_train[0]), "item": np.arange(_num_in_train[1])}
        else:   # train_group_neg_shared.py:24-27
            self.group_sampler = GroupSampler(data_helper.data["train"], group_by="item", chop=conf.chop_size,
                                              device=self.model_train.engine.device.index)
            self.group_sample = self.group_sampler.sample

    def train(self, eval_scheme=None, use_async_eval=True):
        conf = self.conf
        b = conf.batch_size_p
        train = self.data_helper.data["train"]
        engine = self.model_train.engine
        self.train_time = []
        self.mean_n_u = []
        for epoch in range(conf.max_epoch + 1):
            cost, it = 0, 0
            in_window = 0.0
            n_steps = len(train) // b if epoch > 0 else 0
            if self.group_shuffling_trick and getattr(conf, "device_shuffle", False):
                # link list stays in HBM: stratified shuffle by rocPRIM sorts (csrc/shuffle.hip)
                if epoch == 0:
                    self._dev_links = (
                        torch.as_tensor(np.ascontiguousarray(train[:, 0]).astype(np.int32)).to(engine.device),
                        torch.as_tensor(np.ascontiguousarray(train[:, 1]).astype(np.int32)).to(engine.device))
                uid, cid = group_shuffle_train_device(self._dev_links[0], self._dev_links[1], by="item", chop=conf.chop_size,
                                                      seed=getattr(conf, "shuffle_seed", 0) * 1000003 + epoch)
            elif self.group_shuffling_trick:
                train = group_shuffle_train(train, by="item", chop=conf.chop_size, iidx=self._iidx["item"])
                uid, cid = self._upload(train, n_steps)
            elif getattr(conf, "device_shuffle", False):
                # every batch from the GroupSampler (:56), drawn on the device: groups from the B1 device sampler's
                # stream, members by counter-based hash (csrc/shuffle.hip: nncf_group_sample)
                t0 = time.time()
                if n_steps:
                    uid, cid = self.group_sampler.sample_device(b, n_steps, seed=getattr(conf, "shuffle_seed", 0) * 1000003 + epoch)
                    torch.cuda.synchronize(engine.device)
                else:
                    uid = cid = torch.zeros(0, dtype=torch.int32, device=engine.device)
                in_window = time.time() - t0
            else:
                # every batch is drawn from the GroupSampler (:56); the reference draws inside its
                # timed loop, here the epoch's batches are drawn first (same RNG call order: nothing
                # else consumes the streams in between) and their cost is added to the window
                t0 = time.time()
                rows = [self.group_sample(b) for _ in range(n_steps)]
                in_window = time.time() - t0
                rows = np.concatenate(rows, 0) if rows else np.zeros((0, 3), dtype=np.int64)
                uid, cid = self._upload(rows, n_steps)
            n_u = torch.zeros(max(n_steps, 1), dtype=torch.int32, device=engine.device)
            torch.cuda.synchronize(engine.device)
            t_start = time.time()
            if n_steps:
                losses = self.model_train.run_epoch(uid, cid, n_steps, n_u_out=n_u)
                cost = float(losses.double().sum().item())
                it = self._iterations(n_steps, len(train), b)
            if epoch > 0:
                self.train_time.append(time.time() - t_start + in_window)
                self.mean_n_u.append(float(n_u[:n_steps].double().mean().item()))
            self._epoch_report(epoch, it, cost, eval_scheme)
        self.data_helper.data["train"] = train
        self._finish()
