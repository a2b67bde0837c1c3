"""models/train_original.py:8-88 -- plain negative sampling.

Per epoch: np.random.shuffle(train) on the host exactly as the reference (:36; train_base.shuffle_rows), then the whole
`while` loop (:43-68) runs on the device inside nncf_epoch_original: the epoch's negatives are
drawn first (the same LCG stream consumed in the same order: step t uses draws
[t*b*k, (t+1)*b*k) == the reference's per-step sample_batch(k*b) calls, :51), batches are
assembled implicitly (positives, then k negatives per positive, labels by position, :49-56),
and the per-step costs come back in one transfer.  The timing window is the reference's
(:42,69-70): in-loop sampling included, shuffle and evaluation excluded."""
import time

import numpy as np
import torch

from .train_base import TrainerBase, shuffle_rows


class Trainer(TrainerBase):
    def __init__(self, model_dict, conf, data_helper):
        super(Trainer, self).__init__(model_dict, conf, data_helper)
        self.model_train = model_dict["model"]
        self.model_all_loss = model_dict["model_all_loss"]
        self.neg_sign = np.array([-1], dtype="int32") if conf.loss == "skip-gram" else 0  # :13-14
        self.sample_batch = data_helper.sampler_dict["sample_batch"]
        self.monitor_rate = getattr(conf, "monitor_rate_all_loss", 0)   # :16-19: > 0 to monitor the other losses

    def train(self, eval_scheme=None, use_async_eval=True):
        conf = self.conf
        k, b = conf.num_negatives, conf.batch_size_p
        train = self.data_helper.data["train"]
        engine = self.model_train.engine
        sampler = self.sample_batch.__self__
        self.train_time = []
        for epoch in range(conf.max_epoch + 1):
            shuffle_rows(train)
            cost, it = 0, 0
            monitor = (dict((n, 0.0) for n in self.model_all_loss.keys()), 0)
            n_steps = len(train) // b if epoch > 0 else 0  # ragged tail dropped (:46-48)
            uid, cid = self._upload(train, n_steps)
            torch.cuda.synchronize(engine.device)
            t_start = time.time()
            if n_steps:
                neg = sampler.sample_batch_device(n_steps * b * k)
                losses, mon_costs, mon_it = self._run_epoch_monitored(self.model_train, self.model_all_loss, "original", uid, cid,
                                                                      n_steps, neg=neg)   # :59-67
                monitor = (mon_costs, mon_it)
                cost = float(losses.double().sum().item())  # one host sync per epoch
                it = self._iterations(n_steps, len(train), b)
            if epoch > 0:
                self.train_time.append(time.time() - t_start)
            self._epoch_report(epoch, it, cost, eval_scheme, monitor)
        self._finish()
