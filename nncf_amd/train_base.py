"""models/train_base.py:6-41"""
import numpy as np
import torch

from .objectives import Evaluator, get_cur_time  # noqa: F401


def shuffle_rows(a):
    """np.random.shuffle(a) for a 2-D link array, in place: the same permutation and the same RNG
    consumption (permutation(n) is shuffle(arange(n)), and a row shuffle draws the same swap
    indices), a fancy-indexed copy instead of n row swaps -- 7x faster on 200K links, and this call
    sits in every epoch of every trainer (train_original.py:36)."""
    a[:] = a[np.random.permutation(len(a))]


def nan_detection(val_name, val):
    """utils/utilities.py:33-35"""
    if np.isnan(val):
        assert False, "ERROR (nan)! {} is {}..".format(val_name, val)


class TrainerBase(object):
    def __init__(self, model_dict, conf, data_helper):
        self.model_dict = model_dict
        self.conf = conf
        self.data_helper = data_helper
        self.data_spec = data_helper.data_spec
        self.model_predict = model_dict["model_neg_shared"]
        self.evaluater = Evaluator(data_helper, self.data_spec, conf)
        self.train_time = []
        self.history = []  # (epoch, it, mean cost, train_results, test_results)
        print("[INFO] Timestamps below are recorded at the end of training/evaluation respectively")

    def test(self, eval_scheme, predict_only=False, use_async_eval=False):
        if eval_scheme == "given":
            return self.evaluater.run(self.model_dict, eval_scheme=eval_scheme, predict_only=predict_only,
                                      use_async_eval=use_async_eval)
        elif eval_scheme == "whole":
            return self.evaluater.run(self.model_predict, eval_scheme=eval_scheme, predict_only=predict_only,
                                      use_async_eval=use_async_eval)

    def predict(self, eval_scheme, pred_saveto=None):
        result = self.test(eval_scheme, predict_only=True)
        if pred_saveto is not None:
            import pickle
            with open(pred_saveto, "wb") as fp:
                pickle.dump(result, fp, 2)
        return result

    # -- shared epoch scaffolding (train_original.py:34-88 and siblings) ----------------------
    def _upload(self, train, n_steps):
        """(N,3) int64 host links -> two int32 device arrays holding the first n_steps batches."""
        e = self.model_predict.engine
        n = n_steps * self.conf.batch_size_p
        uid = torch.as_tensor(np.ascontiguousarray(train[:n, 0]).astype(np.int32)).to(e.device)
        cid = torch.as_tensor(np.ascontiguousarray(train[:n, 1]).astype(np.int32)).to(e.device)
        return uid, cid

    @staticmethod
    def _iterations(n_steps, n_rows, rows_per_step):
        """`it` as the reference's loops count it (train_original.py:43-48 and every sibling): the counter is
        bumped before the ragged-tail check, so an epoch whose link count is not a multiple of the batch
        reports one iteration more than it trained -- and divides the printed cost by it."""
        return n_steps + (1 if n_steps and n_rows % rows_per_step else 0)

    def _epoch_report(self, epoch, it, cost, eval_scheme):
        # train_original.py:69-87
        mean_cost = cost / it if it > 0 else -1
        print(get_cur_time(), "epoch %d (%d it)" % (epoch, it), "cost %.5f" % mean_cost, end=" ")
        nan_detection("cost", cost)
        results = (None, None)
        if eval_scheme is None:
            print("")
        else:
            results = self.test(eval_scheme, use_async_eval=False)
        self.history.append((epoch, it, mean_cost) + tuple(results))

    def _finish(self):
        print("Training time (sec) per epoch:", np.mean(self.train_time) if self.train_time else float("nan"))
