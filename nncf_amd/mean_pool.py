"""Item text tower of model_choice = basic_embedding: the reference's MeanPool
(modules/content/mean_pool.py:46-110) -- word Embedding (+ row dropout) -> AverageEmbeddings
(divides by L always, :27-33) -> Dense(item_dim) -> BatchNormalization over the batch's unique
items -> ReLU -> Dropout(0) -- with explicit forward / backward / SGD update (no autograd).

Everything runs in csrc/tower.hip behind the nncf_tower_* C ABI (gather-mean + Dense in one
kernel, BatchNorm/ReLU, MFMA backward, per-word sparse SGD); torch only holds the parameter
buffers.
[K-mem] Keras-1 defaults: Embedding init uniform(+-0.05), Dense init glorot_uniform with zero
bias, BatchNormalization eps 1e-3, momentum 0.99, biased batch variance, running "std" slot
holds the variance.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


def postings_schedule(pptr, hot_words, unit=512, per_wg=8, max_postings=2048):
    """Static schedule of the word-table backward over the transposed item CSR (nncf_tower_set_postings): every word is cut
    into units of <= `unit` consecutive postings (a word that would need more than `per_wg` of them: `per_wg` equal units),
    all units of a word go to ONE workgroup, a workgroup takes words in id order while it has unit slots left and <=
    `max_postings` postings (one word alone may exceed that).  Hot words keep a slot (the optimiser visits every word) with
    an empty range: their gradient rows come from the Dense backward.  Returns ([n_wg * per_wg][4] int32, n_wg)."""
    hot = set(int(w) for w in hot_words)
    rows, cur, cur_post = [], [], 0

    def flush():
        while len(cur) < per_wg:
            cur.append((-1, 0, 0, 0))
        rows.extend(cur)
        del cur[:]

    for w in range(len(pptr) - 1):
        b, e = int(pptr[w]), int(pptr[w + 1])
        if w in hot:
            e = b
        df = e - b
        k = max(1, -(-df // unit))
        step = unit
        if k > per_wg:
            k, step = per_wg, -(-df // per_wg)
        if cur and (len(cur) + k > per_wg or cur_post + df > max_postings):
            flush()
            cur_post = 0
        for i in range(k):
            cur.append((w, min(b + i * step, e), min(b + (i + 1) * step, e), k if i == 0 else 0))
        cur_post += df
    if cur:
        flush()
    units = np.asarray(rows, dtype=np.int32).reshape(-1, 4)
    return np.ascontiguousarray(units), len(rows) // per_wg


def build_item_csr(C):
    """Static per-item bag of words of the content matrix C (items x L):
    ptr[items+1], word[nnz] (ascending inside an item), cnt[nnz] (uint8, saturating at 255)."""
    C = np.asarray(C, dtype=np.int64)
    n_items, L = C.shape
    wmax = int(C.max()) + 1
    key = np.repeat(np.arange(n_items, dtype=np.int64), L) * wmax + C.reshape(-1)
    uk, cnt = np.unique(key, return_counts=True)
    items, words = uk // wmax, uk % wmax
    ptr = np.zeros(n_items + 1, dtype=np.int64)
    np.add.at(ptr, items + 1, 1)
    assert cnt.max() <= 255 or L > 255, "token count overflow"
    return np.cumsum(ptr).astype(np.int32), words.astype(np.int32), np.minimum(cnt, 255).astype(np.uint8)


class MeanPoolTower(object):
    """Parameters live in torch tensors (so tests / checkpoints can read them); the native tower
    object (nncf_tower_*) borrows their pointers and owns every scratch buffer."""

    def __init__(self, C, word_count, word_dim, item_dim, learn_rate, device, seed=0, dropout_rate=0.0,
                 eps=1e-3, momentum=0.99, max_rows=512, conv=None, dense=True):
        """conv: None (MeanPool) or dict(filters=F, filter_len=fl, dropout=p): the CNN tower of
        cnn_embedding (modules/content/cnn_model.py) -- one Convolution1D('same') + BatchNorm + relu +
        Dropout + average pooling in front of the same Dense/BN/ReLU tail.
        dense=False: no item_dense_transform (mean_pool.py:78-101, the else branch): the pooled block is the item row (its
        width -- word_dim, or the conv filters -- must be item_dim); the Dense / BatchNorm tensors exist but are never used."""
        self.device = device
        self.dense = bool(dense)
        self._L = _lib.lib()
        C = np.ascontiguousarray(np.asarray(C))
        self.item_count, self.Lc = C.shape
        assert self.Lc <= 255, "max_content_len above 255 needs wider token counts"
        # logical widths as configured, physical widths padded to a multiple of 4 with zero columns /
        # rows (see Engine: zero pads get zero gradients from every layer, BatchNorm maps an all-zero
        # channel to zero, so the padded model is exactly the configured one)
        pad4 = lambda v: (int(v) + 3) // 4 * 4  # noqa: E731
        self.word_count = int(word_count)
        self.wd_l, self.d_l = int(word_dim), int(item_dim)
        self.wd, self.d = pad4(word_dim), pad4(item_dim)
        assert self.wd <= 128 and self.d <= 128, "word_dim / item_dim above 128 are not built"
        self.lr = float(learn_rate)
        self.eps, self.mom = float(eps), float(momentum)
        self.dropout = float(dropout_rate)
        self.cap = (int(max_rows) + 7) // 8 * 8
        gen = torch.Generator().manual_seed(int(seed) + 17)
        dev = device
        W = torch.zeros(self.word_count, self.wd)
        W[:, :self.wd_l] = (torch.rand(self.word_count, self.wd_l, generator=gen) * 2 - 1) * 0.05
        self.W = W.to(dev).contiguous()
        self.conv = dict(conv) if conv else None
        # filter bank (cnn_model.py:63-81): filter_lens = [1, 3, 5] -> three Convolution1D of `filters` filters each,
        # concatenated along the filter axis; bank g owns columns [g F, g F + F_l) of the padded pooled block
        self.filter_lens = [int(v) for v in (self.conv.get("filter_lens") or [self.conv["filter_len"]])] if self.conv else []
        self.nb = len(self.filter_lens)
        self.F_l = int(self.conv["filters"]) if self.conv else 0
        self.F = pad4(self.F_l)
        self.hd_l = self.nb * self.F_l if self.conv else self.wd_l     # Dense input width
        self.hd = self.nb * self.F if self.conv else pad4(self.hd_l)
        lim = float(np.sqrt(6.0 / (self.hd_l + self.d_l)))  # glorot_uniform
        Wd = torch.zeros(self.hd, self.d)
        Wd0 = (torch.rand(self.hd_l, self.d_l, generator=gen) * 2 - 1) * lim
        Wd[self.dense_rows(), :self.d_l] = Wd0
        self.Wd = Wd.to(dev).contiguous()
        # one allocation for the five small vectors, each 512-byte aligned
        vec = torch.zeros(5, 128, device=dev)
        self.bd, self.gamma, self.beta, self.rm, self.rv = (vec[i, :self.d] for i in range(5))
        self.gamma.fill_(1.0)
        self.rv.fill_(1.0)
        self._vec = vec
        self.C = torch.as_tensor(C.astype(np.int32)).to(dev).contiguous()
        ptr, words, cnt = build_item_csr(C)
        self.item_ptr = torch.as_tensor(ptr).to(dev)
        self.item_word = torch.as_tensor(words).to(dev)
        self.item_cnt = torch.as_tensor(cnt).to(dev)
        self.version = 0  # bumped by every update (the evaluator caches inference embeddings)
        extra = [None] * 8
        F = fl = 0
        cdrop = 0.0
        if self.conv:
            F, fl, cdrop = self.F, self.filter_lens[0], float(self.conv.get("dropout", 0.0))
            assert F <= 128 and all(v % 2 == 1 for v in self.filter_lens) and 1 <= self.nb <= 4, \
                "more than 128 filters per length / even filter lengths / more than 4 lengths are not built"
            # Convolution1D init='he_normal' (cnn_model.py:88): std = sqrt(2 / fan_in), fan_in = fl * word_dim [K-mem]
            # one tensor for all banks: [sum fl][wd][F], bank g = taps [tap_off[g], tap_off[g] + fl_g)
            self.tap_off = [int(v) for v in np.concatenate([[0], np.cumsum(self.filter_lens)])]
            Wc = torch.zeros(self.tap_off[-1], self.wd, F)
            for g, flg in enumerate(self.filter_lens):
                Wc[self.tap_off[g]:self.tap_off[g + 1], :self.wd_l, :self.F_l] = \
                    torch.randn(flg, self.wd_l, self.F_l, generator=gen) * float(np.sqrt(2.0 / (flg * self.wd_l)))
            self.Wc = Wc.to(dev).contiguous()
            vec1 = torch.zeros(5, self.nb, 128, device=dev)
            if self.nb == 1:
                self.bc, self.g1, self.b1, self.rm1, self.rv1 = (vec1[i, 0, :F] for i in range(5))
            else:   # one row per bank
                self.bc, self.g1, self.b1, self.rm1, self.rv1 = (vec1[i, :, :F] for i in range(5))
            self.g1.fill_(1.0)
            self.rv1.fill_(1.0)
            self._vec1 = vec1
            # static index of every item's tokens: positions sorted by (word, position), and for each
            # CSR entry (item, word) where its positions start in that order
            spos = np.argsort(C, axis=1, kind="stable").astype(np.uint8)
            cnt64 = cnt.astype(np.int64)
            csum = np.cumsum(cnt64) - cnt64                       # exclusive prefix over all entries
            starts = (csum - np.repeat(csum[ptr[:-1]], np.diff(ptr))).astype(np.uint8)
            self.item_spos = torch.as_tensor(spos).to(dev).contiguous()
            self.item_start = torch.as_tensor(starts).to(dev)
            extra = [self.Wc, self.bc, self.g1, self.b1, self.rm1, self.rv1, self.item_spos, self.item_start]
        cfg = _lib.TowerCfg(self.word_count, self.wd, self.d, self.Lc, self.cap, self.item_count, self.eps, self.mom,
                            self.lr, F, fl, cdrop)
        if self.nb > 1:
            cfg.n_filter_lens = self.nb
            for g, flg in enumerate(self.filter_lens):
                cfg.filter_lens[g] = flg
        prm = _lib.TowerParams(*([t.data_ptr() for t in (self.W, self.Wd, self.bd, self.gamma, self.beta, self.rm,
                                                         self.rv, self.C, self.item_ptr, self.item_word, self.item_cnt)] +
                                 [(t.data_ptr() if t is not None else None) for t in extra]))
        h = ctypes.c_void_p()
        idx = torch.device(dev).index          # 'cuda' without an index means the CURRENT device (as Communicator.rccl resolves it)
        if idx is None:
            idx = torch.cuda.current_device()
        _lib.check(self._L.nncf_tower_create(ctypes.byref(cfg), ctypes.byref(prm), idx, ctypes.byref(h)))
        self.handle = h
        if not self.dense:   # before hot words / postings: both lean on the Dense backward's workgroups
            _lib.check(self._L.nncf_tower_set_no_dense(self.handle, 1))
        if self.dense and not self.conv and os.environ.get("NNCF_HOT_WORDS", "1") != "0":
            # the words that sit in most items (the pad id in all of them): their gradient rows are summed by the Dense
            # backward's workgroups instead of one wave of the word kernel (nncf_tower_set_hot_words); speed only
            df = np.bincount(words, minlength=self.word_count)          # document frequency: one CSR entry per (item, word)
            top = np.argsort(-df, kind="stable")[:16]
            top = np.ascontiguousarray(top[df[top] * 8 >= self.item_count].astype(np.int32))   # in >= 1/8 of the items
            self.hot_words = top
            if top.size:
                _lib.check(self._L.nncf_tower_set_hot_words(self.handle, top.ctypes.data_as(_lib.c_i32p), int(top.size)))
        if self.dense and not self.conv and self.cap > 1024 and self.item_count <= (1 << 24) and os.environ.get("NNCF_WORD_POSTINGS", "1") != "0":
            # blocks of thousands of rows (`original`, the row-wise schemes): the word kernel walks the static transposed CSR
            # (word -> items that hold it) instead of scanning a ~1 % full count matrix (nncf_tower_set_postings); speed only
            order = np.argsort(words, kind="stable")                       # entries by word, items ascending inside a word
            item_of = np.repeat(np.arange(self.item_count, dtype=np.int64), np.diff(ptr))
            pptr = np.zeros(self.word_count + 1, dtype=np.int64)
            np.cumsum(np.bincount(words, minlength=self.word_count), out=pptr[1:])
            packed = (item_of[order] | (cnt[order].astype(np.int64) << 24)).astype(np.uint32)   # item | count << 24
            units, n_wg = postings_schedule(pptr, getattr(self, "hot_words", ()))
            self.post_units = torch.as_tensor(units).to(dev)
            self.post_item = torch.as_tensor(packed.view(np.int32)).to(dev)
            self.slot_of = torch.zeros(self.item_count, dtype=torch.int64, device=dev)   # {stamp, block row} per item
            _lib.check(self._L.nncf_tower_set_postings(self.handle, self.post_units.data_ptr(), n_wg, self.post_item.data_ptr(),
                                                       self.slot_of.data_ptr()))
        if self.dropout > 0 or cdrop > 0:
            self.set_dropout(self.dropout, int(seed) + 29)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self._L.nncf_tower_destroy(h)

    def _stream(self):
        return _lib.stream_ptr(torch.cuda.current_stream(self.device))

    def dense_rows(self):
        """rows of the padded Dense weight matrix that are real inputs, in the configured order (bank after bank)"""
        if not self.conv:
            return torch.arange(self.hd_l)
        return torch.cat([torch.arange(g * self.F, g * self.F + self.F_l) for g in range(self.nb)])

    def bank_Wc(self, g):
        """view of bank g's filters [fl_g][wd][F] (padded widths)"""
        return self.Wc[self.tap_off[g]:self.tap_off[g + 1]]

    def set_params(self, **kw):
        """W, Wd, bd, gamma, beta (+ Wc, bc, g1, b1 for the CNN tower) from host arrays."""
        for name, v in kw.items():
            if v is not None:
                getattr(self, name).copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))
        if self.conv and "Wc" in kw:
            _lib.check(self._L.nncf_tower_sync_params(self.handle, self._stream()))
        self.version += 1

    def conv_mask(self, step_index, m):
        """Dropout(conv_dropout) mask of the step_index-th train-mode forward, first m rows (checkers)."""
        out = torch.empty(self.nb, m, self.Lc, self.F, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_conv_mask(self.handle, int(step_index), m, ctypes.c_void_p(out.data_ptr()),
                                                    self._stream()))
        return out[0] if self.nb == 1 else out   # a filter bank: one (m, L, F) mask per bank

    def set_optimizer(self, kind, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """'sgd' | 'sparse_adam' | 'adam' for the tower's parameters (see Engine.set_optimizer)."""
        mode = {"sgd": 0, "sparse_adam": 1, "adam": 2}[kind]
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        self.opt_state = None
        if mode:
            nd = self.hd * self.d + 384
            if self.conv:
                nd += self.Wc.numel() + 384 * self.nb   # per bank: [fl][wd][F] filters | bias, BN scale, BN shift at 128-float pitch
            z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=self.device)  # noqa: E731
            self.opt_state = {"mW": z(self.word_count, self.wd), "vW": z(self.word_count, self.wd), "mD": z(nd), "vD": z(nd)}
        st = self.opt_state or {}
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_optimizer(self.handle, mode, beta1, beta2, epsilon, p(st.get("mW")),
                                                        p(st.get("vW")), p(st.get("mD")), p(st.get("vD"))))
        self.optimizer = kind
        if getattr(self, "Cid", None) is not None:
            self._bind_content_id(kind)

    def set_pooling(self, pooling):
        """Mean-pool tower: conf.pooling (mean_pool.py:68-76): 'average' (default) | 'max' (MaxPooling1D over the whole
        length).  CNN tower: conf.poolings[0] (cnn_model.py:100-109): how the conv output is pooled over the positions."""
        assert pooling in ("average", "max"), "[ERROR] unknown pooling: %s" % pooling
        self.pooling = pooling
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_pooling(self.handle, int(pooling == "max")))
        self.version += 1

    DENSE_ACTIVATIONS = ("relu", "tanh", "sigmoid", "hard_sigmoid", "softplus", "softsign", "linear", "softmax")

    def set_dense_tail(self, activation="relu", dropout=0.0):
        """item_dense_transform['dense_hidden_actv'] / ['dense_hidden_dropout'] (mean_pool.py:82-97, cnn_model.py:114-126)."""
        if activation not in self.DENSE_ACTIVATIONS:
            raise NotImplementedError("dense_hidden_actv = %r: built are %s" % (activation, ", ".join(self.DENSE_ACTIVATIONS)))
        if self.d_l != self.d and activation in ("sigmoid", "hard_sigmoid", "softplus", "softmax"):
            raise NotImplementedError("dense_hidden_actv = %r needs item_dim %% 4 == 0 (act(0) != 0 would fill the pad columns)" % activation)
        self.dense_act, self.dense_dropout = activation, float(dropout)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_dense_tail(self.handle, self.DENSE_ACTIVATIONS.index(activation), self.dense_dropout))
        self.version += 1

    def set_no_bn(self, no_bn=True):
        """conf.no_BN (mean_pool.py:88-93): Dense -> Activation -> Dropout without the BatchNormalization."""
        self.no_bn = bool(no_bn)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_no_bn(self.handle, int(self.no_bn)))
        self.version += 1

    def dense_mask(self, step_index, m):
        """Dropout(dense_hidden_dropout) mask of the step_index-th train-mode forward, first m rows (checkers)."""
        out = torch.empty(m, self.d, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_dense_mask(self.handle, int(step_index), m, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out[:, :self.d_l] if hasattr(self, "d_l") else out

    def set_c_reg(self, c_reg):
        """conf.c_reg (mean_pool.py:56-59, cnn_model.py:45-48): activity_l2(c_reg) on the word Embedding's output."""
        self.c_reg = float(c_reg)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_c_reg(self.handle, self.c_reg))

    def set_content_id(self, v_reg=0.0, seed=0, optimizer="sgd"):
        """conf.use_content_id (mean_pool.py:102-108): the tower's output + Emb_Cid[item id]; Embedding(item_count,
        item_dim) init uniform(+-0.05) [K-mem], activity_l2(v_reg) on its output."""
        gen = torch.Generator().manual_seed(int(seed) + 41)
        Cid = torch.zeros(self.item_count, self.d)
        Cid[:, :self.d_l] = (torch.rand(self.item_count, self.d_l, generator=gen) * 2 - 1) * 0.05
        self.Cid = Cid.to(self.device).contiguous()
        self.v_reg = float(v_reg)
        self._bind_content_id(optimizer)
        self.version += 1

    def _bind_content_id(self, optimizer):
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        st = {}
        if optimizer != "sgd":
            st = dict(m=torch.zeros_like(self.Cid), v=torch.zeros_like(self.Cid))
            if optimizer == "adam":
                st["stamp"] = torch.zeros(self.item_count, dtype=torch.int32, device=self.device)
        self.cid_state = st
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_set_content_id(self.handle, p(self.Cid), self.v_reg, p(st.get("m")), p(st.get("v")),
                                                         p(st.get("stamp"))))

    def set_dropout(self, rate, seed=0):
        """Embedding(dropout=rate) row dropout (mean_pool.py:56-59), mask generated inside the
        kernels from (seed, train-forward index, word)."""
        self.dropout = float(rate)
        _lib.check(self._L.nncf_tower_set_dropout(self.handle, self.dropout, int(seed)))

    def dropout_scale(self, step_index):
        """per-word scale of the step_index-th train-mode forward since set_dropout (checkers)."""
        out = torch.empty(self.word_count, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_dropout_scale(self.handle, int(step_index), ctypes.c_void_p(out.data_ptr()),
                                                        self._stream()))
        return out

    def forward(self, item_ids, train=True, n_valid=None, out=None, prepare_backward=True, row_scale=None):
        """item embeddings (m, item_dim) for int32 device ids.  train=True: batch statistics over
        the first *n_valid rows (a device int32[1], default all m; rows past it come out zero),
        running statistics updated, and (prepare_backward) the state backward_update needs.
        train=False: inference phase (running statistics), any m."""
        m = item_ids.numel()
        assert item_ids.dtype == torch.int32 and item_ids.is_contiguous()
        E = out if out is not None else torch.empty(m, self.d, dtype=torch.float32, device=self.device)
        scale = row_scale if train else None   # explicit mask (kept alive until the backward ran)
        self._scale = scale
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_forward(self.handle, p(item_ids), m, p(n_valid), int(bool(train)),
                                                  int(bool(train and prepare_backward)), p(scale), p(E), self._stream()))
        return E

    def backward_update(self, dA):
        assert dA.is_contiguous() and dA.dtype == torch.float32
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_backward(self.handle, ctypes.c_void_p(dA.data_ptr()), self._stream()))
        self.version += 1
