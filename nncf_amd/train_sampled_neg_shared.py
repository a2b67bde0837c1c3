"""models/train_sampled_neg_shared.py:8-67 -- k sampled negatives shared by the whole batch.
Per epoch the links are stratified by item (group_shuffle_train, :32-33); every step draws k
negatives (sample_batch(k), :47 -- here one device draw of n_steps*k, the same stream in the same
order) and the inner loop (:37-52) runs on the device."""
import time

import numpy as np
import torch

from .data_utils import group_shuffle_train
from .train_base import TrainerBase


class Trainer(TrainerBase):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        self.model_train = model_dict["model_sampled_neg_shared"]
        self.sample_batch = data_helper.sampler_dict["sample_batch"]
        _num_in_train = np.max(data_helper.data["train"], axis=0) + 1
        self._iidx = {"user": np.arange(_num_in_train[0]), "item": np.arange(_num_in_train[1])}

    def train(self, eval_scheme=None, use_async_eval=True):
        conf = self.conf
        b, k = conf.batch_size_p, conf.num_negatives
        train = self.data_helper.data["train"]
        engine = self.model_train.engine
        sampler = self.sample_batch.__self__
        self.train_time = []
        for epoch in range(conf.max_epoch + 1):
            train = group_shuffle_train(train, by="item", chop=conf.chop_size, iidx=self._iidx["item"])
            cost, it = 0, 0
            n_steps = len(train) // b if epoch > 0 else 0
            uid, cid = self._upload(train, n_steps)
            torch.cuda.synchronize(engine.device)
            t_start = time.time()
            if n_steps:
                neg = sampler.sample_batch_device(n_steps * k)
                losses = self.model_train.run_epoch(uid, cid, n_steps, neg_shared_all=neg)
                cost = float(losses.double().sum().item())
                it = self._iterations(n_steps, len(train), b)
            if epoch > 0:
                self.train_time.append(time.time() - t_start)
            self._epoch_report(epoch, it, cost, eval_scheme)
        self.data_helper.data["train"] = train
        self._finish()
