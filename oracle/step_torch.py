"""TEST / BASELINE INFRASTRUCTURE ONLY -- torch-CPU restatement of one training step of the
reference path, written the way the reference itself runs it: a framework graph (gathers, dense
ops, the loss as the code of utils/objectives.py writes it) differentiated by autograd on the host
cores.  This is bench.py's `cpu_baseline` leg ("the build's CPU restatement
of the full step, torch-CPU, torch.set_num_threads(nproc) and = 1") -- the Keras-1.2.2 / TF-1.0
reference itself cannot run here (section 8(c)).  Only tests/ and bench.py's cpu_baseline leg may
import this module; nncf_amd never does.  tests/test_oracle_torch.py pins it to the numpy oracle
(oracle/step_oracle.py), which tests/test_loss_golden.py pins to the reference's own text.

Follows (all under /root/reference):
  models/model_framework.py:45-48,59-63,85-111,118-136   gathers, tf.unique, normalisation, heads
  modules/content/mean_pool.py:27-33,46-110               MeanPool tower (Embedding row dropout given
                                                          as a per-word scale vector)
  modules/interaction/interaction_dot.py:92-107           mul / matmul
  utils/objectives.py:35-117,163-220                      the three loss families x four losses
  utils/utilities.py:122-135                              activity_l2
Optimiser: the north-star's sparse SGD (touched rows only), as the numpy oracle.
"""
import numpy as np
import torch


def _l2n(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))   # tf.nn.l2_normalize [K-mem]


def _nls(x):
    return -torch.log(torch.sigmoid(x))


class TorchStep(object):
    """U (users, d); V (items, d) id table or None with `tower` = dict(C, W, Wd, bd, gamma, beta)."""

    def __init__(self, U, V, conf, tower=None, eps=1e-3, momentum=0.99, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.conf = conf
        f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        self.U = f(U)
        self.V = f(V) if V is not None else None
        self.tw = None
        if tower is not None:
            self.C = torch.as_tensor(np.asarray(tower["C"]).astype(np.int64))
            self.tw = {k: f(tower[k]) for k in ("W", "Wd", "bd", "gamma", "beta")}
            self.rm, self.rv = torch.zeros_like(self.tw["beta"]), torch.ones_like(self.tw["beta"])
        self.eps, self.mom = eps, momentum
        self.row_scale = None
        self.lam = float(np.float32(conf.neg_loss_weight))     # objectives.py:23,32: np.float32 constants
        self.gamma = float(np.float32(conf.loss_gamma))

    # -- item side -------------------------------------------------------------------------
    def _items(self, ids, leaves):
        """rows of the item block for `ids` (the tower runs on tf.unique(ids), model_framework.py:45-48,98-106)"""
        if self.tw is None:
            V = self.V.detach().requires_grad_(True)
            leaves["V"] = V
            out = V[ids]
        else:
            uniq, inv = torch.unique(ids, return_inverse=True)     # order is irrelevant to the arithmetic
            p = {k: v.detach().requires_grad_(True) for k, v in self.tw.items()}
            leaves.update(p)
            toks = self.C[uniq]
            E = p["W"][toks]
            if self.row_scale is not None:
                E = E * torch.as_tensor(self.row_scale)[toks].unsqueeze(-1)
            h = E.sum(1) / toks.shape[1]                           # mean_pool.py:30: divides by L always
            z = h @ p["Wd"] + p["bd"]
            mu, var = z.mean(0), z.var(0, unbiased=False)
            y = torch.relu(p["gamma"] * (z - mu) * torch.rsqrt(var + self.eps) + p["beta"])
            with torch.no_grad():
                self.rm = self.mom * self.rm + (1 - self.mom) * mu
                self.rv = self.mom * self.rv + (1 - self.mom) * var
            out = y[inv]
        return _l2n(out) if self.conf.normalize_item else out

    def _finish(self, loss, leaves, U, uid, item_ids):
        loss.backward()
        lr = float(self.conf.learn_rate)
        with torch.no_grad():
            tu = torch.unique(uid)
            self.U[tu] -= lr * U.grad[tu]
            if self.tw is None:
                tv = torch.unique(item_ids)
                self.V[tv] -= lr * leaves["V"].grad[tv]
            else:
                for k in self.tw:
                    self.tw[k] -= lr * leaves[k].grad
        return float(loss.item())

    def _users(self, uid):
        U = self.U.detach().requires_grad_(True)
        e = U[uid]
        reg = float(self.conf.u_reg) * (e * e).mean(0).sum()       # utilities.py:122-135
        return U, (_l2n(e) if self.conf.normalize_user else e), reg

    # -- the three heads -------------------------------------------------------------------
    def step_neg_shared(self, uid, cid):
        uid, cid = torch.as_tensor(np.asarray(uid)).long(), torch.as_tensor(np.asarray(cid)).long()
        b, loss = len(uid), self.conf.loss
        leaves = {}
        U, u, reg = self._users(uid)
        v = self._items(cid, leaves)
        pred = u @ v.t()                                            # interaction_dot.py:100-103
        eye = torch.eye(b)
        if loss == "max-margin":
            task = torch.relu((self.gamma - self.gamma * eye) - (torch.diagonal(pred) - pred)).mean()
        elif loss == "log-loss":
            task = _nls(self.gamma * (torch.diagonal(pred) - pred)).mean()
        else:
            wv = float(np.float32(self.conf.neg_loss_weight) / np.float32(b - 1))
            w = eye * (1 - wv) + wv
            task = ((w * _nls((2 * eye - 1) * pred)).sum() if loss == "skip-gram" else (w * (pred - eye) ** 2).sum()) / b
        return self._finish(task + reg, leaves, U, uid, cid)

    def step_group_neg_shared(self, uid, cid):
        uid, cid = torch.as_tensor(np.asarray(uid)).long(), torch.as_tensor(np.asarray(cid)).long()
        b, loss = len(uid), self.conf.loss
        cid_u, cid_x = torch.unique(cid, return_inverse=True)
        leaves = {}
        U, u, reg = self._users(uid)
        v = self._items(cid_u, leaves)
        pred = u @ v.t()
        nu = pred.shape[1]
        rows = torch.arange(b)
        pos = pred[rows, cid_x].reshape(-1, 1)
        mask = torch.zeros_like(pred)
        mask[rows, cid_x] = 1
        if loss == "max-margin":
            task = torch.relu(self.gamma - (pos - pred)).mean()
        elif loss == "log-loss":
            task = _nls(self.gamma * (pos - pred)).mean()
        else:
            wv = self.lam / (nu - 1)
            w = mask * (1 - wv) + wv
            task = ((w * _nls((2 * mask - 1) * pred)).sum() if loss == "skip-gram" else (w * (pred - mask) ** 2).sum()) / b
        return self._finish(task + reg, leaves, U, uid, cid_u)

    def step_original(self, uid, cid, y):
        uid, cid = torch.as_tensor(np.asarray(uid)).long(), torch.as_tensor(np.asarray(cid)).long()
        y = torch.as_tensor(np.asarray(y), dtype=torch.float32).reshape(-1, 1)
        b, k, loss = self.conf.batch_size_p, self.conf.num_negatives, self.conf.loss
        leaves = {}
        U, u, reg = self._users(uid)
        v = self._items(cid, leaves)
        pred = (u * v).sum(1, keepdim=True)                         # interaction_dot.py:92-95
        if loss in ("max-margin", "log-loss"):
            diff = pred[:b].repeat(1, k).reshape(-1, 1) - pred[b:]
            task = torch.relu(self.gamma - diff).mean() if loss == "max-margin" else _nls(self.gamma * diff).mean()
        else:
            wv = float(np.float32(self.conf.neg_loss_weight) / np.float32(k))
            if loss == "skip-gram":
                task = ((1 + (1 - y) / 2.0 * (wv - 1)) * _nls(y * pred)).sum() / b
            else:
                task = ((1 + (1 - y) * (wv - 1)) * (y - pred) ** 2).sum() / b
        return self._finish(task + reg, leaves, U, uid, cid)
