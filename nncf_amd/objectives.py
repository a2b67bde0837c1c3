"""Evaluation half of the reference's utils/objectives.py (:231-492), `whole@k` scheme, on the
GPU: chunked U.V^T scores (nncf_predict_scores) + per-user top-k ranking and map / recall /
precision (nncf_eval_topk), instead of a (users x items) host matrix and a Python loop per user.

The loss half of objectives.py (:15-220) lives inside the HIP step kernels
(csrc/step_original.hip, csrc/step_shared.hip).

Scope: eval_scheme 'whole' (BASELINE's Recall@50).  'given' (pair scoring + AUC over
test_seen/test with sampled negatives, :231-294) needs pandas/sklearn-style per-user AUC and is
not part of the measured path; it raises NotImplementedError.
"""
import ctypes
import os
from datetime import datetime

import numpy as np
import torch

from . import _lib


def get_cur_time():
    return datetime.now().strftime("%Y-%m-%d %H:%M:%S")  # utils/utilities.py:19-20


def _truth_csr(links, candidates, user_count):
    """CSR (ptr, idx) of the 0/1 truth matrix the reference builds densely
    (objectives.py:408-416): row = user, column = slot of the item in sorted(candidates)."""
    cand = np.asarray(candidates, dtype=np.int64)
    items = np.asarray(links[:, 1], dtype=np.int64)
    users = np.asarray(links[:, 0], dtype=np.int64)
    slot = np.searchsorted(cand, items)
    ok = (slot < cand.size) & (cand[np.minimum(slot, cand.size - 1)] == items)
    pairs = np.unique(np.stack([users[ok], slot[ok]], 1), axis=0)  # sorted by (user, slot), duplicates once
    ptr = np.zeros(user_count + 1, dtype=np.int64)
    np.add.at(ptr, pairs[:, 0] + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr.astype(np.int32), pairs[:, 1].astype(np.int32)


class Evaluator(object):
    """utils/objectives.py:373-492."""

    def __init__(self, data_helper, data_spec, conf, eval_scheme=None):
        self.data_helper = data_helper
        self.data_spec = data_spec
        self.conf = conf
        self.eval_scheme = eval_scheme
        if eval_scheme is not None:
            self._check_eval_sheme(eval_scheme)
        self.train = data_helper.data["train"]
        self.test_seen = data_helper.data.get("test_seen")
        self.test = data_helper.data["test"]
        self.user_count = data_spec.user_count
        self.item_count = data_spec.item_count
        self.eval_topk = conf.eval_topk
        self._dev = None
        self._prepare_for_whole_eval()

    def _check_eval_sheme(self, eval_scheme):
        assert eval_scheme == "given" or eval_scheme == "whole", "[Error] Unknown eval_scheme {}.".format(eval_scheme)

    def _prepare_for_whole_eval(self):
        # objectives.py:395-421: candidate sets = sorted(set(items)) of train / test
        train, test = self.train, self.test
        self.train_items = np.unique(train[:, 1])
        self.test_items = np.unique(test[:, 1])
        self._train_csr = _truth_csr(train, self.train_items, self.user_count)
        self._test_csr = _truth_csr(test[test[:, 2] == 1], self.test_items, self.user_count)

    # ------------------------------------------------------------------------------------
    def _device_state(self, engine):
        if self._dev is None or self._dev["device"] != engine.device:
            to = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(engine.device)  # noqa: E731
            self._dev = {
                "device": engine.device,
                "users": torch.arange(self.user_count, dtype=torch.int32, device=engine.device),
                "train": (to(self.train_items), to(self._train_csr[0]), to(self._train_csr[1])),
                "test": (to(self.test_items), to(self._test_csr[0]), to(self._test_csr[1])),
                "metrics": torch.zeros(self.user_count * 3, dtype=torch.float32, device=engine.device),
                "valid": torch.zeros(self.user_count, dtype=torch.int32, device=engine.device),
            }
        return self._dev

    FUSED_MAX_TOPK = 100   # nncf_eval_whole (csrc/eval.hip kEvalMaxK)

    def _whole(self, model_predict, which, topk, user_chunk=2048, fused=None):
        """test_eval_mat + evaluate_mat (objectives.py:296-370) for one candidate set.  Default: ONE kernel scores and ranks
        (nncf_eval_whole: no (users x candidates) matrix); fused=False (or NNCF_EVAL_FUSED=0, or eval_topk > 100) takes the
        two-kernel path over a score chunk -- both leave the same bits."""
        predict_engine = getattr(model_predict, "predict_engine", None)
        engine = predict_engine() if predict_engine is not None else model_predict.engine
        st = self._device_state(engine)
        items, ptr, idx = st[which]
        n_cand = int(items.numel())
        L = _lib.lib()
        metrics, valid = st["metrics"], st["valid"]
        if fused is None:
            fused = os.environ.get("NNCF_EVAL_FUSED", "1") != "0"
        if fused and topk <= self.FUSED_MAX_TOPK and hasattr(engine, "_h"):
            with torch.cuda.device(engine.device):
                stream = _lib.stream_ptr(torch.cuda.current_stream(engine.device))
                _lib.check(L.nncf_eval_whole(
                    engine._h, ctypes.c_void_p(st["users"].data_ptr()), self.user_count, ctypes.c_void_p(items.data_ptr()), n_cand,
                    ctypes.c_void_p(ptr.data_ptr()), ctypes.c_void_p(idx.data_ptr()), int(topk),
                    ctypes.c_void_p(metrics.data_ptr()), ctypes.c_void_p(valid.data_ptr()), stream))
            return self._mean(metrics, valid, topk)
        scores = torch.empty((min(user_chunk, self.user_count), n_cand), dtype=torch.float32, device=engine.device)
        with torch.cuda.device(engine.device):
            stream = _lib.stream_ptr(torch.cuda.current_stream(engine.device))
            for u0 in range(0, self.user_count, user_chunk):
                u1 = min(self.user_count, u0 + user_chunk)
                model_predict.predict_on_batch_device(st["users"][u0:u1], items, out=scores)
                _lib.check(L.nncf_eval_topk(
                    ctypes.c_void_p(scores.data_ptr()), u1 - u0, n_cand, n_cand,
                    ctypes.c_void_p(ptr[u0:].data_ptr()), ctypes.c_void_p(idx.data_ptr()), int(topk),
                    ctypes.c_void_p(metrics[3 * u0:].data_ptr()), ctypes.c_void_p(valid[u0:].data_ptr()), stream))
        return self._mean(metrics, valid, topk)

    @staticmethod
    def _mean(metrics, valid, topk):
        v = valid.bool()
        m = metrics.view(-1, 3)[v].double()
        n = int(v.sum().item())
        mean = m.mean(0).cpu().numpy() if n else np.zeros(3)
        return {"map@%d" % topk: float(mean[0]), "recall@%d" % topk: float(mean[1]),
                "precision@%d" % topk: float(mean[2]), "auc": -1.0, "users": n}

    def run(self, model, predict_only=False, verbose=True, eval_scheme=None, batch_size=1024, use_async_eval=False):
        """objectives.py:423-492.  `use_async_eval` is accepted for signature compatibility; the
        ranking already runs asynchronously on the GPU stream and the (tiny) result is fetched
        here, so there is no host thread sharing np.random with the trainer."""
        if eval_scheme is None:
            eval_scheme = self.eval_scheme
        self._check_eval_sheme(eval_scheme)
        eval_topk = self.eval_topk
        if eval_scheme == "given":
            raise NotImplementedError("eval_scheme='given' (objectives.py:231-294, pair AUC) is outside the "
                                      "measured path; use 'whole@k'")
        assert eval_topk > 0, "[ERROR] eval_top {} must > 0".format(eval_topk) + "when eval_scheme=whole"
        if predict_only:
            raise NotImplementedError("predict_only materialises the (users x items) matrices on the host "
                                      "(objectives.py:317-318); not built")
        train_results = self._whole(model, "train", eval_topk)
        test_results = self._whole(model, "test", eval_topk)
        if verbose:
            print(get_cur_time(), "train recall/map", train_results["recall@%s" % eval_topk],
                  train_results["map@%s" % eval_topk], "test recall/map", test_results["recall@%s" % eval_topk],
                  test_results["map@%s" % eval_topk])
        return train_results, test_results
