#!/usr/bin/env python3
"""tests/golden/loss_golden.npz: outputs of the REFERENCE's own loss / interaction / regulariser
code, executed in this container from where it lies under /root/reference.

The Keras-1.2.2 / TF-1.0 half of the reference cannot be imported (py2 syntax, packages absent),
but the arithmetic of rows a11, a15-a20 of SURVEY.md section 8 is plain Python over a handful of
backend primitives:
    utils/objectives.py:15-220              _get_neg_loss_weight, _get_gamma, get_original_loss,
                                            get_neg_shared_loss, get_sampled_neg_shared_loss,
                                            get_group_neg_shared_loss
    modules/interaction/interaction_dot.py:10-22,82-107   normalize_shape, InteractionDot.call
    utils/utilities.py:67-83,122-135        tensorflow_reshape, tensorflow_diag, L1L2Act
Those source lines are read at run time, the two-line py2 `print` statements inside them are
replaced by `pass` in memory, and they are exec'd against a primitive-op `K` / `tf` pair written
on torch-CPU (class T below: one thin tensor wrapper + ~25 elementwise / indexing primitives,
each the documented behaviour of the TF-1.0 / Keras-1 call of the same name).  Autograd of that
execution gives d(loss)/d(scores) and d(loss)/d(table rows).  Nothing of the reference (source,
bytecode) is written anywhere: the fixture holds inputs, seeds and the produced numbers only.

What stays [K-mem] (taken from memory of the absent packages, implemented in the shim, not in
the exec'd text): K.repeat's (samples, n, dim) layout, tf.unique's first-occurrence order,
tf.nn.l2_normalize's epsilon (1e-12 under max), K.relu'(0) = 0, Embedding lookup = row gather
whose gradient sums duplicate rows, and Keras' train_on_batch = mean(loss tensor) + activity
regularisers of the layers on the graph.

Every case is evaluated twice: in float32 (what the reference computes) and in float64 (the
formulas to rounding error, the tight pin for the oracle).

    python3 tools/gen_golden_losses.py        # rewrites tests/golden/loss_golden.npz
"""
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_integrity  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_inputs import case_inputs  # noqa: E402
LOSSES = ("skip-gram", "mse", "log-loss", "max-margin")


# ---------------------------------------------------------------------------------------------
# primitive-op backend on torch-CPU
# ---------------------------------------------------------------------------------------------
class T(object):
    """A tensor with TF's operator conventions: numpy arrays / Python scalars on either side are
    converted to the tensor operand's dtype (tf.convert_to_tensor with a dtype hint)."""
    __array_ufunc__ = None  # numpy defers `ndarray op T` to T.__rop__

    def __init__(self, t):
        self.t = t

    @staticmethod
    def lift(x, like):
        if isinstance(x, T):
            return x.t
        return torch.as_tensor(np.asarray(x), dtype=None).to(like.dtype) if not isinstance(x, torch.Tensor) else x

    def _bin(self, other, f, rev=False):
        o = T.lift(other, self.t)
        if self.t.dtype.is_floating_point and not o.dtype.is_floating_point:
            o = o.to(self.t.dtype)
        return T(f(o, self.t) if rev else f(self.t, o))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.div)
    def __rtruediv__(self, o): return self._bin(o, torch.div, True)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return T(-self.t)
    def __pow__(self, p): return T(self.t ** p)

    def __getitem__(self, idx):
        return T(self.t[idx])

    @property
    def _keras_shape(self):
        return tuple(self.t.shape)

    def _shape_as_list(self):
        return list(self.t.shape)


def _idx(i):
    i = i.t if isinstance(i, T) else torch.as_tensor(np.asarray(i))
    return i.long()


class KBackend(object):
    """keras.backend of Keras 1.2.2 (TensorFlow backend): only what the exec'd lines call."""
    _backend = "tensorflow"

    def __init__(self, dtype):
        self.dtype = dtype

    def repeat(self, x, n):   # [K-mem] (samples, dim) -> (samples, n, dim)
        return T(x.t.unsqueeze(1).repeat(1, int(n), 1))

    def mean(self, x, axis=None):
        return T(x.t.mean() if axis is None else x.t.mean(dim=axis))

    def sum(self, x, axis=None):
        return T(x.t.sum() if axis is None else x.t.sum(dim=axis))

    def relu(self, x): return T(torch.relu(x.t))          # [K-mem] gradient 0 at 0
    def log(self, x): return T(torch.log(x.t))
    def sigmoid(self, x): return T(torch.sigmoid(x.t))
    def abs(self, x): return T(torch.abs(x.t))
    def square(self, x): return T(x.t * x.t)

    def cast_to_floatx(self, x):
        return np.asarray(x, dtype=np.float64 if self.dtype == torch.float64 else np.float32)


class TFBackend(object):
    """tensorflow 1.0: only what the exec'd lines call."""
    float32 = "float32"

    def __init__(self, dtype):
        self.dtype = dtype
        self.nn = self

    def _t(self, x):
        if isinstance(x, T):
            return x.t
        a = np.asarray(x)
        t = torch.as_tensor(a)
        return t.to(self.dtype) if a.dtype.kind == "f" else t

    def constant(self, x): return T(self._t(x))
    def Variable(self, x, name=None): return T(self._t(x))
    def assign(self, var, value): return T(self._t(value).clone())
    def reshape(self, x, shape): return T(self._t(x).reshape(tuple(int(s) for s in shape)))
    def transpose(self, x): return T(self._t(x).t())
    def matmul(self, a, b): return T(self._t(a) @ self._t(b))
    def diag_part(self, x): return T(torch.diagonal(self._t(x)))

    def reduce_sum(self, x, axis=None, keep_dims=False):
        return T(self._t(x).sum(dim=axis, keepdim=keep_dims))

    def gather(self, params, ids):
        return T(self._t(params)[_idx(ids)])

    def gather_nd(self, params, indices):
        p, i = self._t(params), _idx(indices)
        return T(p[tuple(i[..., c] for c in range(i.shape[-1]))])

    def scatter_nd_add(self, ref, indices, updates):
        p, i = self._t(ref), _idx(indices)
        return T(p.index_put(tuple(i[..., c] for c in range(i.shape[-1])), self._t(updates), accumulate=True))

    def shape(self, x):
        return T(torch.as_tensor(list(self._t(x).shape), dtype=torch.int64))

    def cast(self, x, dtype):
        return T(self._t(x).to(self.dtype))

    def slice(self, x, begin, size):
        size = [int(s) for s in (size.t.tolist() if isinstance(size, T) else size)]
        t = self._t(x)
        return T(t[begin[0]:begin[0] + size[0], begin[1]:begin[1] + size[1]])

    def concat(self, xs, axis):
        return T(torch.cat([self._t(x) for x in xs], dim=axis))

    def unique(self, x):   # [K-mem] values in first-occurrence order, index of every element
        v = self._t(x).reshape(-1).tolist()
        seen, uniq, inv = {}, [], []
        for e in v:
            if e not in seen:
                seen[e] = len(uniq)
                uniq.append(e)
            inv.append(seen[e])
        return T(torch.as_tensor(uniq, dtype=torch.int64)), T(torch.as_tensor(inv, dtype=torch.int64))

    def l2_normalize(self, x, dim=-1, epsilon=1e-12):   # [K-mem] x * rsqrt(max(sum x^2, eps))
        t = self._t(x)
        ss = (t * t).sum(dim=dim, keepdim=True)
        return T(t * torch.rsqrt(torch.clamp(ss, min=epsilon)))


# ---------------------------------------------------------------------------------------------
# the reference's text, exec'd from where it lies
# ---------------------------------------------------------------------------------------------
def _lines(path):
    return ref_integrity.lines(path)        # sha256-checked against the reviewed revision before anything is exec'd


def _strip_py2_prints(src):
    """`print '...' \\` + continuation line (py2 statements inside except-branches) -> pass"""
    src = re.sub(r"print '[^\n]*\\\n[^\n]*", "pass", src)      # two-line statements first
    return re.sub(r"print ['\"][^\n]*", "pass", src)


def reference_namespace(dtype):
    K, tf = KBackend(dtype), TFBackend(dtype)
    ns = {"np": np, "K": K, "tf": tf, "Regularizer": object}
    # utils/utilities.py:67-83 tensorflow_reshape / tensorflow_diag, :122-135 L1L2Act
    ul = _lines("utils/utilities.py")
    a = next(i for i, l in enumerate(ul) if l.startswith("def tensorflow_reshape"))
    b = next(i for i, l in enumerate(ul) if l.startswith("def theano_diag"))
    c = next(i for i, l in enumerate(ul) if l.startswith("def tensorflow_diag"))
    d = next(i for i, l in enumerate(ul) if l.startswith("def theano_get_shape"))
    e = next(i for i, l in enumerate(ul) if l.startswith("class L1L2Act"))
    f = next(i for i, l in enumerate(ul) if i > e and l.startswith("    def get_config"))
    exec(compile("\n".join(ul[a:b] + ul[c:d] + ul[e:f]), "ref_utilities", "exec"), ns)
    K.reshape = ns["tensorflow_reshape"]     # utilities.py:110-113 (tensorflow backend branch)
    K.diag = ns["tensorflow_diag"]
    # utils/objectives.py:15-220
    ol = _lines("utils/objectives.py")
    a = next(i for i, l in enumerate(ol) if l.startswith("def _get_neg_loss_weight"))
    b = next(i for i, l in enumerate(ol) if l.startswith("#        Evaluations"))
    exec(compile(_strip_py2_prints("\n".join(ol[a:b - 1])), "ref_objectives", "exec"), ns)
    # modules/interaction/interaction_dot.py:10-22 normalize_shape, :82-107 InteractionDot.call
    il = _lines("modules/interaction/interaction_dot.py")
    a = next(i for i, l in enumerate(il) if l.startswith("def normalize_shape"))
    b = next(i for i, l in enumerate(il) if l.startswith("class InteractionDot"))
    c = next(i for i, l in enumerate(il) if l.startswith("    def call(self"))
    d = next(i for i, l in enumerate(il) if l.startswith("    def set_form"))
    call_src = "\n".join(l[4:] if l.startswith("    ") else l for l in il[c:d])
    exec(compile("\n".join(il[a:b]) + "\n" + call_src, "ref_interaction", "exec"), ns)
    return ns


class Interact(object):
    """the attribute bag InteractionDot.call reads (interaction_dot.py:83-107)"""

    def __init__(self, call, bias, ubias, cbias):
        self._call, self.bias, self.ubias, self.cbias, self.form = call, bias, ubias, cbias, "mul"

    def set_form(self, form):
        self.form = form
        return self

    def __call__(self, x):
        return self._call(self, x)


class Conf(object):
    pass


# ---------------------------------------------------------------------------------------------
# one case = model_framework.py:59-63,85-88,109-149 wired by hand around the exec'd pieces
# ---------------------------------------------------------------------------------------------
def run_case(ns, dtype, family, loss, U, V, uid, cid, b, k, lam, gamma, u_reg, normalize, mf, bias, ub, cb):
    tf = ns["tf"]
    conf = Conf()
    conf.neg_loss_weight, conf.loss_gamma = lam, gamma
    leaf = lambda a: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)  # noqa: E731
    Ut, Vt = leaf(U), leaf(V)
    ubt = leaf(ub.reshape(-1, 1)) if bias in ("user", "both") else None
    cbt = leaf(cb.reshape(-1, 1)) if bias in ("item", "both") else None
    uid_t = torch.as_tensor(uid.reshape(-1, 1).astype(np.int64))
    cid_t = torch.as_tensor(cid.reshape(-1, 1).astype(np.int64))
    # Emb_U(uid) with activity_l2(conf.u_reg) on its output (model_framework.py:59-61; utilities.py:137-138)
    U_emb_raw = T(Ut[uid_t.reshape(-1)])
    reg = ns["L1L2Act"](l2=u_reg)(U_emb_raw) if u_reg else 0.0
    U_emb = tf.nn.l2_normalize(U_emb_raw, dim=-1) if normalize else U_emb_raw
    inter = Interact(ns["call"], bias, T(ubt) if ubt is not None else None, T(cbt) if cbt is not None else None)
    norm_item = normalize and not mf    # model_framework.py:109-111 sits in the content-model branch only
    extra = {}
    if family == "group_neg_shared":
        cid_u, cid_x = tf.unique(T(cid_t.reshape(-1)))
        C_compact = T(Vt[cid_u.t])
        if norm_item:
            C_compact = tf.nn.l2_normalize(C_compact, dim=-1)
        pred = inter.set_form("matmul")([U_emb, C_compact, T(uid_t), T(cid_u.t.reshape(-1, 1))])
        row_prefix = tf.Variable(np.arange(b, dtype="int32").reshape((b, 1)))          # model_framework.py:38-39
        pos_idxs = tf.concat([row_prefix, tf.reshape(cid_x, (-1, 1))], 1)             # :133-134
        pred.t.retain_grad()
        lt = ns["get_group_neg_shared_loss"](pred, pos_idxs, loss, b, conf)
        extra = {"cid_u": cid_u.t.numpy().astype(np.int32), "cid_x": cid_x.t.numpy().astype(np.int32)}
    else:
        C_emb = T(Vt[cid_t.reshape(-1)])
        if norm_item:
            C_emb = tf.nn.l2_normalize(C_emb, dim=-1)
        if family == "original":
            pred = inter.set_form("mul")([U_emb, C_emb, T(uid_t), T(cid_t)])
            y = np.ones((b * (1 + k), 1))
            y[b:] = -1 if loss == "skip-gram" else 0                                  # train_original.py:13-14
            pred.t.retain_grad()
            lt = ns["get_original_loss"](loss, b, k, conf)(T(torch.as_tensor(y, dtype=dtype)), pred)
        elif family == "neg_shared":
            pred = inter.set_form("matmul")([U_emb, C_emb, T(uid_t), T(cid_t)])
            pred.t.retain_grad()
            lt = ns["get_neg_shared_loss"](loss, b, conf)(None, pred)
        else:   # sampled_neg_shared, model_framework.py:64-65,114-143: U_emb_front is NOT normalised
            U_front = T(Ut[uid_t.reshape(-1)[:b]])
            pos = inter.set_form("mul")([U_front, C_emb[:b], T(uid_t[:b]), T(cid_t[:b])])
            neg = inter.set_form("matmul")([U_front, C_emb[b:], T(uid_t[:b]), T(cid_t[b:])])
            pred = tf.concat([pos, neg], 1)
            pred.t.retain_grad()
            lt = ns["get_sampled_neg_shared_loss"](loss, b, k, conf)(None, pred)
            if u_reg:   # the head's graph holds Emb_U twice: on uid (U_emb, unused) and on uid_front
                reg = ns["L1L2Act"](l2=u_reg)(U_front)
    task = lt.t.mean()                      # Keras: mean of the loss tensor ("task_loss * loss_base")
    total = task + (reg.t if isinstance(reg, T) else reg)
    total.backward()
    out = {"loss": total.item(), "task_loss": task.item(), "pred": pred.t.detach().numpy().copy(),
           "dpred": pred.t.grad.numpy().copy(), "dU": Ut.grad.numpy().copy(), "dV": Vt.grad.numpy().copy()}
    if ubt is not None:
        out["dub"] = ubt.grad.numpy().reshape(-1).copy()
    if cbt is not None:
        out["dcb"] = cbt.grad.numpy().reshape(-1).copy()
    out.update(extra)
    return out


FAMILIES = ("original", "neg_shared", "group_neg_shared", "sampled_neg_shared")


def main():
    ns32, ns64 = reference_namespace(torch.float32), reference_namespace(torch.float64)
    out, meta = {}, []
    R = np.random.RandomState(99).randn(128, 3)     # projection for the large cases
    out["proj"] = R
    case = 0
    for family in FAMILIES:
        for loss in LOSSES:
            lam_demo, gamma_demo = (8. if loss == "mse" else 128.), (0.1 if loss == "max-margin" else 10.)
            pair = loss in ("log-loss", "max-margin")
            variants = [
                # (tag, b, k, d, nu, ni, scale, lam, gamma, u_reg, normalize, mf, bias, full)
                ("toy", 8, 2, 4, 12, 10, 0.8, 16., gamma_demo if not pair else 2.0, 1e-2, pair, True, None, True),
                ("toy_content", 8, 2, 4, 12, 10, 0.8, 16., 2.0 if pair else gamma_demo, 1e-2, True, False, None, True),
                ("toy_bias", 8, 2, 4, 12, 10, 0.8, 16., 2.0 if pair else gamma_demo, 0.0, pair, True, "both", True),
                ("b32", 32, 3, 12, 40, 30, 0.5, 16., gamma_demo, 1e-3, pair, True, None, True),
                ("b32_bias", 32, 3, 12, 40, 30, 0.5, 16., gamma_demo, 1e-3, pair, False, "both", True),
                ("b512", 512, 10, 100, 700, 600, 0.17, lam_demo, gamma_demo, 1e-6, pair, True, None, False),
                ("b512_content", 512, 10, 100, 700, 600, 0.17, lam_demo, gamma_demo, 1e-6, pair, False, None, False),
            ]
            for (tag, b, k, d, nu, ni, scale, lam, gamma, u_reg, normalize, mf, bias, full) in variants:
                seed = 1000 + case
                U, V, uid, cid, ub, cb = case_inputs(seed, family, b, k, d, nu, ni, scale)
                r32 = run_case(ns32, torch.float32, family, loss, U, V, uid, cid, b, k, lam, gamma, u_reg, normalize, mf, bias, ub, cb)
                r64 = run_case(ns64, torch.float64, family, loss, U, V, uid, cid, b, k, lam, gamma, u_reg, normalize, mf, bias, ub, cb)
                assert np.isfinite(r32["loss"]) and np.isfinite(r64["loss"]), (family, loss, tag)
                p = "c%d_" % case
                meta.append("%s|%s|%s|%d|%d|%d|%d|%d|%r|%r|%r|%r|%d|%d|%s|%d|%d" % (
                    family, loss, tag, b, k, d, nu, ni, scale, lam, gamma, u_reg, int(normalize), int(mf), bias or "", int(full), seed))
                out[p + "uid"], out[p + "cid"] = uid.astype(np.int32), cid.astype(np.int32)
                if full:
                    out[p + "U"], out[p + "V"] = U, V
                if bias:
                    out[p + "ub"], out[p + "cb"] = ub, cb
                for nm, r, ft in (("32", r32, np.float32), ("64", r64, np.float64)):
                    out[p + "loss" + nm] = np.array([r["loss"], r["task_loss"]], dtype=np.float64)
                    if full:
                        for key in ("pred", "dpred", "dU", "dV"):
                            out[p + key + nm] = r[key].astype(ft)
                    else:   # large: rows 0..7 in full + a 3-column random projection of everything
                        for key in ("dU", "dV"):
                            out[p + key + "_head" + nm] = r[key][:8].astype(ft)
                            if nm == "64":
                                out[p + key + "_proj" + nm] = (r[key].astype(np.float64) @ R[:d]).astype(np.float64)
                        if nm == "64":
                            out[p + "dpred_rowsum" + nm] = r["dpred"].astype(np.float64).sum(axis=1)
                            out[p + "dpred_colsum" + nm] = r["dpred"].astype(np.float64).sum(axis=0)
                    for key in ("dub", "dcb"):
                        if key in r:
                            out[p + key + nm] = r[key].astype(ft)
                if "cid_u" in r64:
                    out[p + "cid_u"], out[p + "cid_x"] = r64["cid_u"], r64["cid_x"]
                case += 1
    out["meta"] = np.array(meta)
    path = os.path.join(ROOT, "tests", "golden", "loss_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB,", case, "cases")


if __name__ == "__main__":
    sys.exit(main())
