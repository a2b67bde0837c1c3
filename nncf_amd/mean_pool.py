"""Item text tower of model_choice = basic_embedding: the reference's MeanPool
(modules/content/mean_pool.py:46-110) -- word Embedding (+ row dropout) -> AverageEmbeddings
(divides by L always, :27-33) -> Dense(item_dim) -> BatchNormalization over the batch's unique
items -> ReLU -> Dropout(0) -- with explicit forward / backward / SGD update (no autograd).

Split of work: the memory-bound halves (gather + mean of n_u*L word rows forward, sparse
word-table update backward) are HIP kernels (csrc/meanpool.hip); Dense / BatchNorm / ReLU on the
pooled (n_u x word_dim) block are a few library GEMMs and elementwise ops through torch (the
plain-library-GEMM allowance).
[K-mem] Keras-1 defaults: Embedding init uniform(+-0.05), Dense init glorot_uniform with zero
bias, BatchNormalization eps 1e-3, momentum 0.99, biased batch variance, running "std" slot
holds the variance.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def build_inverted_index(C, word_count):
    """word -> postings (item ascending, count) of the content matrix C (items x L)."""
    C = np.asarray(C, dtype=np.int64)
    items = np.repeat(np.arange(C.shape[0], dtype=np.int64), C.shape[1])
    key = C.reshape(-1) * C.shape[0] + items
    uk, cnt = np.unique(key, return_counts=True)
    words, its = uk // C.shape[0], uk % C.shape[0]
    ptr = np.zeros(word_count + 1, dtype=np.int64)
    np.add.at(ptr, words + 1, 1)
    return np.cumsum(ptr).astype(np.int32), its.astype(np.int32), cnt.astype(np.int32)


class MeanPoolTower(object):
    def __init__(self, C, word_count, word_dim, item_dim, learn_rate, device, seed=0, dropout_rate=0.0,
                 eps=1e-3, momentum=0.99):
        self.device = device
        self._L = _lib.lib()
        C = np.ascontiguousarray(np.asarray(C))
        self.item_count, self.Lc = C.shape
        self.word_count, self.wd, self.d = int(word_count), int(word_dim), int(item_dim)
        assert self.wd % 4 == 0 and self.wd <= 128, "word_dim must be a multiple of 4 and <= 128"
        self.lr = float(learn_rate)
        self.eps, self.mom = float(eps), float(momentum)
        self.dropout = float(dropout_rate)
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
        ptr, its, cnt = build_inverted_index(C, self.word_count)
        self.post_ptr = torch.as_tensor(ptr).to(dev)
        self.post_item = torch.as_tensor(its).to(dev)
        self.post_cnt = torch.as_tensor(cnt).to(dev)
        self.slot_of_item = torch.full((self.item_count,), -1, dtype=torch.int32, device=dev)
        self._mask_gen = torch.Generator(device=dev).manual_seed(int(seed) + 29)
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

    def forward(self, item_ids, train=True):
        """item embeddings (n, item_dim) for int32 device ids; train=True caches for backward."""
        scale = None
        if train and self.dropout > 0:  # Embedding(dropout=p): whole word rows dropped per batch (:56-59)
            keep = 1.0 - self.dropout
            scale = (torch.rand(self.word_count, generator=self._mask_gen, device=self.device) < keep).float() / keep
        h = self._pool(item_ids, scale)
        z = torch.addmm(self.bd, h, self.Wd)
        if train:
            mu = z.mean(0)
            var = ((z - mu) ** 2).mean(0)
        else:
            mu, var = self.rm, self.rv
        inv = torch.rsqrt(var + self.eps)
        xhat = (z - mu) * inv
        y = self.gamma * xhat + self.beta
        if train:
            self._cache = (item_ids, h, xhat, inv, y, scale, mu, var)
        return torch.relu(y)

    def backward_update(self, dA):
        item_ids, h, xhat, inv, y, scale, mu, var = self._cache
        n = float(dA.shape[0])
        dy = dA * (y > 0)
        dgamma = (dy * xhat).sum(0)
        dbeta = dy.sum(0)
        dxhat = dy * self.gamma
        dz = inv / n * (n * dxhat - dxhat.sum(0) - xhat * (dxhat * xhat).sum(0))
        dWd = h.t() @ dz
        dbd = dz.sum(0)
        dh = (dz @ self.Wd.t()).contiguous()
        # sparse SGD on the word table through the inverted index (fixed order, no atomics)
        ids = item_ids.long()
        self.slot_of_item[ids] = torch.arange(ids.numel(), dtype=torch.int32, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.nncf_meanpool_bwd(p(self.W), self.wd, self.wd, self.word_count, p(self.post_ptr),
                                                 p(self.post_item), p(self.post_cnt), p(self.slot_of_item), p(dh),
                                                 p(scale), self.lr, self.Lc, self._stream()))
        self.slot_of_item[ids] = -1
        self.Wd -= self.lr * dWd
        self.bd -= self.lr * dbd
        self.gamma -= self.lr * dgamma
        self.beta -= self.lr * dbeta
        self.rm = self.mom * self.rm + (1 - self.mom) * mu
        self.rv = self.mom * self.rv + (1 - self.mom) * var
        self.version += 1
