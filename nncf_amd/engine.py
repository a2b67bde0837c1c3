"""Thin Python handle over the B2 entry points of libnncf_hip.so: owns the embedding tables
(torch CUDA tensors = device memory plumbing) and forwards every step to the HIP kernels.
No arithmetic of the hot path happens in Python or torch."""
import ctypes

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()


class Engine(object):
    """One parameter set (user table, item table) + one loss/hyper-parameter configuration.

    Several engines may share tables (`share=other`), which is how the reference's
    loss-monitor models (SGD(0.) clones, model_framework.py:179-193) are expressed."""

    def __init__(self, user_count, item_count, dim, batch_size_p, num_negatives, loss, loss_gamma,
                 neg_loss_weight, u_reg, learn_rate, normalize_user, normalize_item, device=0, seed=0,
                 share=None, init_scale=0.05, tables=None, interaction_bias=None, bias_item_count=None, bias_state=None):
        if loss not in _lib.LOSS_CODES:
            raise AssertionError("[ERROR!] loss %s not specified." % loss)  # objectives.py:72
        self._L = _lib.lib()
        self.device = torch.device("cuda", device)
        # the kernels move rows as 16-byte chunks: a dimension that is not a multiple of 4 (the
        # reference's default is 50) is zero-padded to the next one.  Pad columns start at zero, get a
        # zero gradient from every term of every loss (dot products, norms and the regulariser ignore
        # them) and therefore stay zero under SGD and Adam: the model is exactly the dim-wide one.
        self.dim = int(dim)                       # logical width (what the reference calls user_dim)
        self.dim_p = (self.dim + 3) // 4 * 4      # physical row width of the tables
        self.b = int(batch_size_p)
        self.k = int(num_negatives)
        self.user_count, self.item_count = int(user_count), int(item_count)
        if tables is not None:
            self.U, self.V = tables  # caller-provided device tensors (e.g. shared U, tower output buffer)
            assert self.U.is_cuda and self.V.is_cuda and self.U.is_contiguous() and self.V.is_contiguous()
            assert self.U.shape[1] == self.dim_p and self.V.shape[1] == self.dim_p, "tables must be dim_p wide"
        elif share is None:
            # Keras-1 Embedding default init: uniform(-0.05, 0.05) [K-mem]
            if (self.user_count + self.item_count) * self.dim > 2e8:
                # multi-GB tables (config 4): initialise on the device
                gen = torch.Generator(device=self.device).manual_seed(int(seed))
                def mk(r):
                    t = torch.zeros(r, self.dim_p, device=self.device)
                    t[:, :self.dim] = (torch.rand(r, self.dim, generator=gen, device=self.device) * 2 - 1) * init_scale
                    return t
                self.U, self.V = mk(self.user_count).contiguous(), mk(self.item_count).contiguous()
            else:
                gen = torch.Generator().manual_seed(int(seed))
                U = torch.zeros(self.user_count, self.dim_p)
                V = torch.zeros(self.item_count, self.dim_p)
                U[:, :self.dim] = (torch.rand(self.user_count, self.dim, generator=gen) * 2 - 1) * init_scale
                V[:, :self.dim] = (torch.rand(self.item_count, self.dim, generator=gen) * 2 - 1) * init_scale
                self.U = U.to(self.device).contiguous()
                self.V = V.to(self.device).contiguous()
        else:
            self.U, self.V = share.U, share.V
        cfg = _lib.ModelCfg(self.b, self.k, self.dim_p, self.dim_p, _lib.LOSS_CODES[loss], int(bool(normalize_user)),
                            int(bool(normalize_item)), float(loss_gamma), float(neg_loss_weight), float(u_reg),
                            float(learn_rate), self.user_count, self.item_count)
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_model_create(ctypes.byref(cfg), _ptr(self.U), _ptr(self.V), self.device.index, 1,
                                                 ctypes.byref(self._h)))
        self._loss1 = torch.zeros(1, dtype=torch.float32, device=self.device)
        # InteractionDot bias (interaction_dot.py:62-71): zero-initialised vectors over users / the
        # catalogue; engines over the same parameters share them (bias_state / share)
        assert interaction_bias in (None, "user", "item", "both"), "ERROR! Unknown interation bias"
        self.interaction_bias = interaction_bias
        self.opt_state, self.optimizer = None, "sgd"
        self.sparse_adam_duplicates = "sum"
        self.bias = None
        if interaction_bias:
            if bias_state is None and share is not None:
                bias_state = getattr(share, "bias", None)
            if bias_state is None:
                n_items = int(bias_item_count if bias_item_count is not None else self.item_count)
                z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)  # noqa: E731
                bias_state = {"ub": z(self.user_count) if interaction_bias in ("user", "both") else None,
                              "cb": z(n_items) if interaction_bias in ("item", "both") else None, "n_items": n_items}
            self.bias = bias_state
            self._bind_bias()

    def _bind_bias(self):
        b = self.bias
        if b is None:
            return
        mode = self.OPT_CODES[self.optimizer]
        if mode and "m_ub" not in b:
            z = lambda t: torch.zeros_like(t) if t is not None else None  # noqa: E731
            b.update(m_ub=z(b["ub"]), v_ub=z(b["ub"]), m_cb=z(b["cb"]), v_cb=z(b["cb"]))
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_model_set_bias(self._h, _ptr(b["ub"]), _ptr(b["cb"]), int(b["n_items"]),
                                                   _ptr(b.get("m_ub")), _ptr(b.get("v_ub")), _ptr(b.get("m_cb")),
                                                   _ptr(b.get("v_cb"))))

    OPT_CODES = {"sgd": 0, "sparse_adam": 1, "adam": 2}

    def set_sparse_adam_duplicates(self, mode):
        """'sum' (default; TF's _apply_sparse_duplicate_indices) | 'as_written' (utils/optimizer.py:108-134 on an
        IndexedSlices with duplicates: moments from the last occurrence, one step per occurrence)."""
        assert mode in ("sum", "as_written"), "[ERROR] sparse_adam_duplicates %r unknown" % (mode,)
        self.sparse_adam_duplicates = mode
        _lib.check(self._L.nncf_model_set_sparse_adam_duplicates(self._h, int(mode == "as_written")))

    def set_optimizer(self, kind, beta1=0.9, beta2=0.999, epsilon=1e-8, state=None):
        """'sgd' | 'sparse_adam' (utils/optimizer.py:108-134) | 'adam' (Keras Adam,
        basic_embedding_conf.py:59-60).  `state`: moment tensors of another engine over the same
        tables (heads of one model share one optimiser, model_framework.py:24)."""
        mode = self.OPT_CODES[kind]
        if mode and state is None:
            z = torch.zeros_like
            state = {"mU": z(self.U), "vU": z(self.U)}
            if getattr(self, "_export", None) is None:
                state.update(mV=z(self.V), vV=z(self.V))
        self.opt_state = state if mode else None
        st = state or {}
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_model_set_optimizer(self._h, mode, beta1, beta2, epsilon, _ptr(st.get("mU")),
                                                        _ptr(st.get("vU")), _ptr(st.get("mV")), _ptr(st.get("vV"))))
        self.optimizer = kind
        self._bind_bias()   # bias moments / stamps follow the optimiser

    # -- helpers -------------------------------------------------------------------------
    def ids(self, x):
        """Host or device integer ids -> contiguous int32 CUDA tensor (Keras casts to int32,
        model_framework.py:37)."""
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x).reshape(-1)).astype(np.int32)).to(self.device)

    def set_tables(self, U=None, V=None):
        if U is not None:
            self.U.copy_(torch.as_tensor(U, dtype=torch.float32))
        if V is not None:
            self.V.copy_(torch.as_tensor(V, dtype=torch.float32))

    def set_hparams(self, learn_rate=None, loss_gamma=None, neg_loss_weight=None, u_reg=None):
        c = self.cfg
        if learn_rate is not None:
            c.learn_rate = learn_rate
        if loss_gamma is not None:
            c.loss_gamma = loss_gamma
        if neg_loss_weight is not None:
            c.neg_loss_weight = neg_loss_weight
        if u_reg is not None:
            c.u_reg = u_reg
        _lib.check(self._L.nncf_model_set_hparams(self._h, c.learn_rate, c.loss_gamma, c.neg_loss_weight, c.u_reg))

    def _stream(self):
        return _lib.stream_ptr(torch.cuda.current_stream(self.device))

    def set_shared_device(self, shared=True):
        """The GPU is shared with another process / rank / stream: the one-launch step never waits inside a launch
        (stage + commit form; same results bit for bit).  See nncf_model_set_shared_device."""
        _lib.check(self._L.nncf_model_set_shared_device(self._h, int(bool(shared))))

    def check(self):
        """Synchronises the stream; raises RuntimeError if an in-launch wait of the one-launch step ever timed out."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_model_check(self._h, self._stream()))

    def clear_error(self):
        """Clears the sticky time-out flag `check` reports; the tables hold a partial step and must be restored by the caller."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_model_clear_error(self._h, self._stream()))

    def set_item_grad_export(self, buf):
        """While `buf` ([rows][dim] fp32) is set, steps leave item rows untouched and write
        dL/d(item row) there instead (text towers above the item block, sharded rows)."""
        self._export = buf
        _lib.check(self._L.nncf_model_set_item_grad_export(self._h, _ptr(buf)))

    # -- single steps (device ids in, device loss out; asynchronous) ----------------------
    def step_original(self, uid, cid, update=True, loss_out=None):
        out = self._loss1 if loss_out is None else loss_out
        assert uid.numel() == self.b and cid.numel() == self.b * (1 + self.k)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_step_original(self._h, _ptr(uid), _ptr(cid), _ptr(out), int(update), self._stream()))
        return out

    def step_rowwise(self, uid, cid, y, update=True, loss_out=None):
        out = self._loss1 if loss_out is None else loss_out
        n = self.b * (1 + self.k)
        assert uid.numel() == n and cid.numel() == n and y.numel() == n
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_step_rowwise(self._h, _ptr(uid), _ptr(cid), _ptr(y), _ptr(out), int(update),
                                                 self._stream()))
        return out

    def epoch_rowwise(self, uid_all, cid_all, y_all, n_steps, loss_out=None):
        if loss_out is None:
            loss_out = torch.empty(max(n_steps, 1), dtype=torch.float32, device=self.device)
        n = self.b * (1 + self.k)
        assert min(uid_all.numel(), cid_all.numel(), y_all.numel()) >= n_steps * n
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_epoch_rowwise(self._h, _ptr(uid_all), _ptr(cid_all), _ptr(y_all), int(n_steps),
                                                  _ptr(loss_out), self._stream()))
        return loss_out[:n_steps]

    def step_neg_shared(self, uid, cid, update=True, loss_out=None):
        out = self._loss1 if loss_out is None else loss_out
        assert uid.numel() == self.b and cid.numel() == self.b
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_step_neg_shared(self._h, _ptr(uid), _ptr(cid), _ptr(out), int(update), self._stream()))
        return out

    def step_group_neg_shared(self, uid, cid, update=True, loss_out=None, want_unique=False):
        out = self._loss1 if loss_out is None else loss_out
        assert uid.numel() == self.b and cid.numel() == self.b
        cu = cx = nu = None
        if want_unique:
            cu = torch.empty(self.b, dtype=torch.int32, device=self.device)
            cx = torch.empty(self.b, dtype=torch.int32, device=self.device)
            nu = torch.empty(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_step_group_neg_shared(self._h, _ptr(uid), _ptr(cid), _ptr(out), int(update),
                                                          _ptr(cu), _ptr(cx), _ptr(nu), self._stream()))
        if want_unique:
            return out, cu, cx, nu
        return out

    # -- whole trainer loops (no host sync inside) ----------------------------------------
    def epoch_original(self, uid_all, cid_all, neg_all, n_steps, loss_out=None):
        if loss_out is None:
            loss_out = torch.empty(max(n_steps, 1), dtype=torch.float32, device=self.device)
        assert uid_all.numel() >= n_steps * self.b and neg_all.numel() >= n_steps * self.b * self.k
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_epoch_original(self._h, _ptr(uid_all), _ptr(cid_all), _ptr(neg_all), int(n_steps),
                                                   _ptr(loss_out), self._stream()))
        return loss_out[:n_steps]

    def epoch_neg_shared(self, uid_all, cid_all, n_steps, loss_out=None):
        if loss_out is None:
            loss_out = torch.empty(max(n_steps, 1), dtype=torch.float32, device=self.device)
        assert uid_all.numel() >= n_steps * self.b and cid_all.numel() >= n_steps * self.b
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_epoch_neg_shared(self._h, _ptr(uid_all), _ptr(cid_all), int(n_steps),
                                                     _ptr(loss_out), self._stream()))
        return loss_out[:n_steps]

    def epoch_group_neg_shared(self, uid_all, cid_all, n_steps, loss_out=None, n_u_out=None):
        if loss_out is None:
            loss_out = torch.empty(max(n_steps, 1), dtype=torch.float32, device=self.device)
        assert uid_all.numel() >= n_steps * self.b and cid_all.numel() >= n_steps * self.b
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_epoch_group_neg_shared(self._h, _ptr(uid_all), _ptr(cid_all), int(n_steps),
                                                           _ptr(loss_out), _ptr(n_u_out), self._stream()))
        return loss_out[:n_steps]

    SCHEMES = {"original": 0, "neg_shared": 1, "group_neg_shared": 2}

    def plan_item_block(self, scheme, uid, cid, uniq_out, n_u_out):
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_plan_item_block(self._h, self.SCHEMES[scheme], _ptr(uid), _ptr(cid), _ptr(uniq_out),
                                                    _ptr(n_u_out), self._stream()))

    def step_planned(self, scheme, uid, update=True, loss_out=None):
        out = self._loss1 if loss_out is None else loss_out
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_step_planned(self._h, self.SCHEMES[scheme], _ptr(uid), _ptr(out), int(update),
                                                 self._stream()))
        return out

    def predict_scores(self, uid, cid, out=None):
        p, q = uid.numel(), cid.numel()
        if out is None:
            out = torch.empty((p, q), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_predict_scores(self._h, _ptr(uid), p, _ptr(cid), q, _ptr(out), self._stream()))
        return out

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.nncf_model_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def profile_begin(kernel, max_launches):
    """Bracket every launch of HIP kernel `kernel` with events (see nncf_profile_begin)."""
    _lib.check(_lib.lib().nncf_profile_begin(kernel.encode(), int(max_launches)))


def profile_end():
    """(mean bracket ms, launches, empty-bracket ms) since profile_begin."""
    ms, n, ov = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(_lib.lib().nncf_profile_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(ov)))
    return ms.value, n.value, ov.value


_plan_ws = {}


def plan_unique(ids):
    """tf.unique of a device id list through the plan kernel: (uniq[n] int32 device, first n_u
    valid; inverse[n] int32 device; n_u python int -- one host sync)."""
    uniq, inv, cnt = _plan_unique_impl(ids)
    return uniq, inv, int(cnt[0].item())


def _plan_unique_impl(ids):
    L = _lib.lib()
    n = ids.numel()
    dev = ids.device
    key = (dev, n)
    if key not in _plan_ws:
        mk = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
        _plan_ws[key] = dict(order=mk(n), seg=mk(n + 1), uniq=mk(n), inv=mk(n), cnt=mk(4), desc=mk(4 * n), ss=mk(n),
                             ch=mk(n // 16 + 2))
    w = _plan_ws[key]
    with torch.cuda.device(dev):
        _lib.check(L.nncf_plan_ids(_ptr(ids), n, _ptr(w["order"]), _ptr(w["seg"]), _ptr(w["uniq"]), _ptr(w["inv"]),
                                   _ptr(w["cnt"]), _ptr(w["desc"]), _ptr(w["ss"]), _ptr(w["ch"]),
                                   _lib.stream_ptr(torch.cuda.current_stream(dev))))
    return w["uniq"], w["inv"], w["cnt"]


def plan_unique_device(ids):
    """As plan_unique, without the host sync: (uniq, inverse, counts) device tensors, counts[0] = n_u."""
    uniq, inv, _ = _plan_unique_impl(ids)
    return uniq, inv, _plan_ws[(ids.device, ids.numel())]["cnt"]


def plan_ids(ids):
    """Duplicate-row plan of one id list (tests): returns (order, seg_start, uniq, inverse, nseg)."""
    L = _lib.lib()
    ids = ids.to(dtype=torch.int32).contiguous()
    n = ids.numel()
    dev = ids.device
    order = torch.empty(n, dtype=torch.int32, device=dev)
    seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    uniq = torch.empty(n, dtype=torch.int32, device=dev)
    inverse = torch.empty(n, dtype=torch.int32, device=dev)
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    desc = torch.empty(4 * n, dtype=torch.int32, device=dev)
    slot_seg = torch.empty(n, dtype=torch.int32, device=dev)
    chunk_start = torch.empty(n // 16 + 2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.nncf_plan_ids(_ptr(ids), n, _ptr(order), _ptr(seg_start), _ptr(uniq), _ptr(inverse), _ptr(counts),
                                   _ptr(desc), _ptr(slot_seg), _ptr(chunk_start),
                                   _lib.stream_ptr(torch.cuda.current_stream(dev))))
    nseg = int(counts[0].item())
    nchunks = int(counts[2].item())
    plan_ids.last_desc = desc[:4 * nseg].cpu().numpy().reshape(nseg, 4)
    plan_ids.last_slot_seg = slot_seg.cpu().numpy()
    plan_ids.last_chunks = chunk_start[:nchunks + 1].cpu().numpy()
    return order.cpu().numpy(), seg_start[:nseg + 1].cpu().numpy(), uniq[:nseg].cpu().numpy(), inverse.cpu().numpy(), nseg
