"""models/train_neg_shared.py:8-78 -- negative sharing (no additional negative sampling).
The inner loop (:46-59) runs on the device inside nncf_epoch_neg_shared."""
import time

import numpy as np
import torch

from .train_base import TrainerBase, shuffle_rows


class Trainer(TrainerBase):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        self.model_train = model_dict["model_neg_shared"]
        self.model_all_loss = model_dict["model_neg_shared_all_loss"]
        self.monitor_rate = getattr(conf, "monitor_rate_all_loss", 0)
        if conf.neg_dist != "unigram":  # :24-26
            print("[WARNING] Only unigram neg_dist is currently supported "
                  "for group_neg_shared training. Set neg_dist = unigram.")

    def train(self, eval_scheme=None, use_async_eval=True):
        conf = self.conf
        b = conf.batch_size_p
        train = self.data_helper.data["train"]
        engine = self.model_train.engine
        self.train_time = []
        for epoch in range(conf.max_epoch + 1):
            cost, it = 0, 0
            monitor = (dict((n, 0.0) for n in self.model_all_loss.keys()), 0)
            shuffle_rows(train)
            n_steps = len(train) // b if epoch > 0 else 0
            uid, cid = self._upload(train, n_steps)
            torch.cuda.synchronize(engine.device)
            t_start = time.time()
            if n_steps:
                losses, mon_costs, mon_it = self._run_epoch_monitored(self.model_train, self.model_all_loss, "neg_shared", uid, cid,
                                                                      n_steps)   # :52-58
                monitor = (mon_costs, mon_it)
                cost = float(losses.double().sum().item())
                it = self._iterations(n_steps, len(train), b)
            if epoch > 0:
                self.train_time.append(time.time() - t_start)
            self._epoch_report(epoch, it, cost, eval_scheme, monitor)
        self._finish()
