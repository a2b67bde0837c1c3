"""Item text tower of model_choice = basic_embedding: the reference's MeanPool
(modules/content/mean_pool.py:46-110) -- word Embedding (+ row dropout) -> AverageEmbeddings
(divides by L always, :27-33) -> Dense(item_dim) -> BatchNormalization over the batch's unique
items -> ReLU -> Dropout(0) -- with explicit forward / backward / SGD update (no autograd).

Split of work: the memory-bound halves (gather + mean of n_u*L word rows forward, sparse
word-table update backward) are HIP kernels (csrc/meanpool.hip); Dense / BatchNorm / ReLU on the
pooled (n_u x word_dim) block are five small fused HIP launches (csrc/tower.hip).  torch only
holds the buffers and does the integer content gather.
[K-mem] Keras-1 defaults: Embedding init uniform(+-0.05), Dense init glorot_uniform with zero
bias, BatchNormalization eps 1e-3, momentum 0.99, biased batch variance, running "std" slot
holds the variance.
"""
import ctypes

import numpy as np
import torch

from . import _lib


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
    def __init__(self, C, word_count, word_dim, item_dim, learn_rate, device, seed=0, dropout_rate=0.0,
                 eps=1e-3, momentum=0.99, max_rows=512):
        self.device = device
        self._L = _lib.lib()
        C = np.ascontiguousarray(np.asarray(C))
        self.item_count, self.Lc = C.shape
        assert self.Lc <= 255, "max_content_len above 255 needs wider token counts"
        self.word_count, self.wd, self.d = int(word_count), int(word_dim), int(item_dim)
        assert self.wd % 4 == 0 and self.wd <= 128, "word_dim must be a multiple of 4 and <= 128"
        self.lr = float(learn_rate)
        self.eps, self.mom = float(eps), float(momentum)
        self.dropout = float(dropout_rate)
        self.cap = int(max_rows)
        gen = torch.Generator().manual_seed(int(seed) + 17)
        dev = device
        self.W = ((torch.rand(self.word_count, self.wd, generator=gen) * 2 - 1) * 0.05).to(dev).contiguous()
        lim = float(np.sqrt(6.0 / (self.wd + self.d)))  # glorot_uniform
        self.Wd = ((torch.rand(self.wd, self.d, generator=gen) * 2 - 1) * lim).to(dev).contiguous()
        self.bd = torch.zeros(self.d, device=dev)
        self.gamma = torch.ones(self.d, device=dev)
        self.beta = torch.zeros(self.d, device=dev)
        self.rm = torch.zeros(self.d, device=dev)
        self.rv = torch.ones(self.d, device=dev)
        self.C = torch.as_tensor(C.astype(np.int32)).to(dev).contiguous()
        ptr, words, cnt = build_item_csr(C)
        self.item_ptr = torch.as_tensor(ptr).to(dev)
        self.item_word = torch.as_tensor(words).to(dev)
        self.item_cnt = torch.as_tensor(cnt).to(dev)
        # scratch of the backward transpose: all-zero between calls (the kernels restore it)
        self.cnt_t = torch.zeros(self.word_count * self.cap, dtype=torch.uint8, device=dev)
        self.touched = torch.zeros(self.word_count, dtype=torch.uint8, device=dev)
        self._mask_gen = torch.Generator(device=dev).manual_seed(int(seed) + 29)
        self._ws = None
        self.version = 0  # bumped by every update (the evaluator caches inference embeddings)

    def _stream(self):
        return _lib.stream_ptr(torch.cuda.current_stream(self.device))

    def set_params(self, W=None, Wd=None, bd=None, gamma=None, beta=None):
        for name, v in (("W", W), ("Wd", Wd), ("bd", bd), ("gamma", gamma), ("beta", beta)):
            if v is not None:
                getattr(self, name).copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))
        self.version += 1

    def _pool(self, item_ids, scale):
        toks = self.C[item_ids.long()].contiguous()                       # (n, L) int32 gather
        h = torch.empty(toks.shape[0], self.wd, dtype=torch.float32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_meanpool_fwd(p(self.W), self.wd, self.wd, p(toks), toks.shape[0], self.Lc, p(scale),
                                                 p(h), self._stream()))
        return h

    def forward(self, item_ids, train=True, n_valid=None, out=None):
        """item embeddings (m, item_dim) for int32 device ids.  Only the first *n_valid rows
        (a device int32[1], default all m) are real: the rest are padding that is computed but
        excluded from the BatchNorm statistics -- shapes stay static, nothing syncs to the host.
        train=True uses batch statistics and keeps what backward_update needs."""
        m = item_ids.numel()
        dev = self.device
        if n_valid is None:
            n_valid = torch.full((1,), m, dtype=torch.int32, device=dev)
        scale = None
        if train and self.dropout > 0:  # Embedding(dropout=p): whole word rows dropped per batch (:56-59)
            keep = 1.0 - self.dropout
            scale = (torch.rand(self.word_count, generator=self._mask_gen, device=dev) < keep).float() / keep
        ids = item_ids  # padding slots past n_valid must hold in-range ids (nncf_plan_item_block writes 0)
        h = self._pool(ids, scale)
        z = torch.empty(m, self.d, dtype=torch.float32, device=dev)
        E = out if out is not None else torch.empty(m, self.d, dtype=torch.float32, device=dev)
        if self._ws is None:
            self._ws = torch.zeros(int(self._L.nncf_tower_ws_floats(self.wd, self.d)), dtype=torch.float32, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(self._L.nncf_tower_fwd(p(h), m, p(n_valid), self.wd, self.d, p(self.Wd), p(self.bd),
                                              p(self.gamma), p(self.beta), p(self.rm), p(self.rv), self.eps, self.mom,
                                              int(bool(train)), p(z), p(E), p(self._ws), self._stream()))
        if train:
            self._cache = (ids, n_valid, h, z, scale)
        return E

    def backward_update(self, dA):
        ids, n_valid, h, z, scale = self._cache
        m = ids.numel()
        assert m <= self.cap and dA.shape[0] == m and dA.is_contiguous()
        dh = torch.empty(m, self.wd, dtype=torch.float32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_tower_bwd(p(h), p(z), p(dA), m, p(n_valid), self.wd, self.d, p(self.Wd), p(self.bd),
                                              p(self.gamma), p(self.beta), self.lr, p(dh), p(self._ws), self._stream()))
            # sparse SGD on the word table (fixed order, no atomics)
            _lib.check(self._L.nncf_meanpool_bwd(p(self.W), self.wd, self.wd, self.word_count, p(ids), p(n_valid),
                                                 self.cap, p(self.item_ptr), p(self.item_word), p(self.item_cnt),
                                                 p(self.cnt_t), p(self.touched), p(dh), p(scale), self.lr, self.Lc,
                                                 self._stream()))
        self.version += 1
