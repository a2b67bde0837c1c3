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


def _monitor_loss(model, kind, uid, cid, neg):
    """loss of ONE batch on a loss-monitor model (model_framework.py:179-193), nothing updated; device ids in, device
    scalar out.  kind: 'original' (cid = the b positives, neg = their b * k negatives) | 'neg_shared'."""
    core = getattr(model, "core", None)
    if kind == "original":
        cid = torch.cat([cid, neg])
    if core is not None:                       # text-tower heads
        return core.step(kind, uid, cid, update=False)
    e = model.engine                           # id-table heads
    return e.step_original(uid, cid, update=False) if kind == "original" else e.step_neg_shared(uid, cid, update=False)


class TrainerBase(object):
    def __init__(self, model_dict, conf, data_helper):
        self.model_dict = model_dict
        self.conf = conf
        self.data_helper = data_helper
        self.data_spec = data_helper.data_spec
        self.model_predict = model_dict["model_neg_shared"]
        self.evaluater = Evaluator(data_helper, self.data_spec, conf)
        self.train_time = []
        self.monitor_rate = 0
        self.monitor_history = []   # (epoch, monitored steps, {loss name: summed cost}) when monitor_rate_all_loss > 0
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

    def _run_epoch_monitored(self, head, all_loss, kind, uid, cid, n_steps, neg=None):
        """The epoch's steps through `head.run_epoch` with the reference's loss monitoring
        (train_original.py:38-40,59-67; train_neg_shared.py:34-36,52-58): after a step, with probability
        monitor_rate_all_loss (`random.random()`, one draw per step in step order, as the reference), every model of
        `all_loss` evaluates its loss on that step's batch -- against the parameters the step just left, without
        touching them (the reference's SGD(0.) models).  The device loop is cut after every monitored step; with the
        rate at 0 it is one native call.  Returns (per-step losses, {loss name: summed monitor cost}, monitored steps).
        Deviation: model_framework.py:179-193 compiles all four monitor models with the ONE loss function of
        conf.loss (its `loss` argument is the already-built function, not the name); here each evaluates the loss it
        is named after -- what the epoch line's `( skip-gram=.. mse=.. )` claims to print."""
        import random
        names = list(all_loss.keys())
        costs = dict((n, 0.0) for n in names)
        rate = self.monitor_rate
        if not rate or not n_steps:
            return head.run_epoch(uid, cid, n_steps, neg_all=neg), costs, 0
        e = head.engine
        b, k = e.b, e.k
        hits = [t for t in range(n_steps) if random.random() < rate]
        parts, t0 = [], 0
        for h in hits + [n_steps]:
            cnt = min(h + 1, n_steps) - t0
            if cnt > 0:
                parts.append(head.run_epoch(uid[t0 * b:], cid[t0 * b:], cnt,
                                            neg_all=None if neg is None else neg[t0 * b * k:]).clone())
                t0 += cnt
            if h < n_steps:      # the monitored step's batch, parameters as that step left them
                ub, cb = uid[h * b:(h + 1) * b], cid[h * b:(h + 1) * b]
                nb = None if neg is None else neg[h * b * k:(h + 1) * b * k]
                for name in names:
                    costs[name] += float(_monitor_loss(all_loss[name], kind, ub, cb, nb).item())
        return torch.cat(parts), costs, len(hits)

    def _epoch_report(self, epoch, it, cost, eval_scheme, monitor=None):
        # train_original.py:69-87
        mean_cost = cost / it if it > 0 else -1
        print(get_cur_time(), "epoch %d (%d it)" % (epoch, it), "cost %.5f" % mean_cost, end=" ")
        nan_detection("cost", cost)
        eng = getattr(self.model_predict, "engine", None)     # a timed-out in-launch wait of the one-launch step (a GPU this
        if eng is not None and hasattr(eng, "check"):          # model does not have to itself) is reported with its cause, not
            eng.check()                                        # only as the NaN it leaves in the loss
        if monitor is not None and self.monitor_rate > 0:      # train_original.py:74-79
            costs, monitor_it = monitor
            print("(", end=" ")
            for name in costs:
                print("{}={}".format(name, costs[name] / monitor_it if monitor_it > 0 else -1), end=" ")
            print(")", end=" ")
            self.monitor_history.append((epoch, monitor_it, dict(costs)))
        results = (None, None)
        if eval_scheme is None:
            print("")
        else:
            results = self.test(eval_scheme, use_async_eval=False)
        self.history.append((epoch, it, mean_cost) + tuple(results))

    def _finish(self):
        print("Training time (sec) per epoch:", np.mean(self.train_time) if self.train_time else float("nan"))
