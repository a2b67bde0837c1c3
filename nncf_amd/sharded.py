"""Multi-GPU neg_shared training: one process per GPU, tables range-sharded, RCCL all-to-all
over xGMI for the cross-shard row exchange (no reference counterpart --
the reference is single-process).

Partitioning
  * users: contiguous id ranges; rank r owns users [r*Us, (r+1)*Us) AND every training link of
    those users, so user rows are always local.
  * items: contiguous id ranges; rank r owns item rows [r*Is, (r+1)*Is) (north-star: "the
    embedding table shards by item-id range").
One global step (all ranks call `step` together, weak scaling: b positives per rank)
  1. each rank de-duplicates its batch's items (device plan, tf.unique order);
  2. all-to-all #1: requested item ids -> owners      (fixed capacity b per peer, -1 padded)
  3. owners gather the rows (HIP), all-to-all #2: rows -> requesters
  4. local neg_shared step (tile + apply kernels) on [local user shard, fetched item rows];
     user rows are updated in place, item gradients are exported
  5. all-to-all #3: item gradients -> owners, which sum them in (rank, slot) order and apply the
     sparse SGD update (deterministic scatter, no atomics).
This is synchronous data-parallel SGD: all ranks' batches see the same pre-step item rows and
the owners apply the sum of the per-rank gradients.  `simulate_global_step` in
tests/test_sharded_gloo.py restates exactly that on one process with the oracle.

All collectives have equal, static split sizes (no host synchronisation to size them); xGMI is
a full mesh, so all-to-all drives every link at once, unlike a ring all-reduce.
The exchange logic is backend-agnostic: `HipBackend` runs the kernels, tests inject a numpy
backend and run the same code over gloo on CPU.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist


def shard_size(count, world):
    return (int(count) + world - 1) // world


def partition_links_by_user(train, user_count, world):
    """Rank r gets the links whose user falls in its range (order preserved)."""
    us = shard_size(user_count, world)
    owner = np.asarray(train[:, 0]) // us
    return [train[owner == r] for r in range(world)]


class HipBackend(object):
    """Compute on this rank's GPU through libnncf_hip.so."""

    def __init__(self, engine_kwargs, local_users, local_items, device, seed=0):
        from . import _lib
        from .engine import Engine, plan_ids  # noqa: F401
        self._lib = _lib
        self.device = torch.device("cuda", device)
        kw = dict(engine_kwargs)
        kw["user_count"] = local_users
        self.b = kw["batch_size_p"]
        self.dim = kw["dim"]
        # the model's "item table" is the buffer of fetched rows (at most b unique items per step)
        kw["item_count"] = self.b
        self.engine = Engine(device=device, seed=seed, **kw)
        self.U = self.engine.U                      # local user shard
        self.Vbuf = self.engine.V                   # fetched item rows [b][d]
        gen = torch.Generator().manual_seed(int(seed) + 1)
        self.V = ((torch.rand(local_items, self.dim, generator=gen) * 2 - 1) * 0.05).to(self.device).contiguous()
        self.lr = float(kw["learn_rate"])
        self._gV = torch.zeros(self.b, self.dim, dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws = None

    def ids(self, x):
        return self.engine.ids(x)

    def _stream(self):
        return self._lib.stream_ptr(torch.cuda.current_stream(self.device))

    def unique(self, cid):
        """tf.unique on device: (uniq[b] padded with -1, inverse[b], n_u tensor)."""
        L = self._lib.lib()
        n = cid.numel()
        if self._ws is None:
            mk = lambda k: torch.empty(k, dtype=torch.int32, device=self.device)  # noqa: E731
            self._ws = dict(order=mk(n), seg=mk(n + 1), uniq=mk(n), inv=mk(n), cnt=mk(4), desc=mk(4 * n), ss=mk(n),
                            ch=mk(n // 16 + 2))
        w = self._ws
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self._lib.check(L.nncf_plan_ids(p(cid), n, p(w["order"]), p(w["seg"]), p(w["uniq"]), p(w["inv"]), p(w["cnt"]),
                                        p(w["desc"]), p(w["ss"]), p(w["ch"]), self._stream()))
        n_u = w["cnt"][0:1]
        uniq = torch.where(torch.arange(n, device=self.device) < n_u, w["uniq"], torch.full_like(w["uniq"], -1))
        return uniq, w["inv"], n_u

    def gather_rows(self, ids):
        """rows of the LOCAL item shard for local ids (zeros where id < 0)."""
        L = self._lib.lib()
        out = torch.empty(ids.numel(), self.dim, dtype=torch.float32, device=self.device)
        self._lib.check(L.nncf_gather_rows(ctypes.c_void_p(self.V.data_ptr()), self.dim, self.dim,
                                           ctypes.c_void_p(ids.data_ptr()), ids.numel(),
                                           ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def local_step(self, uid_local, inverse, vrows):
        """neg_shared step on (local user shard, fetched rows): returns (loss[1], item grads [b][d])."""
        L = self._lib.lib()
        self.Vbuf[:vrows.shape[0]].copy_(vrows)
        self._lib.check(L.nncf_step_neg_shared_export(self.engine._h, ctypes.c_void_p(uid_local.data_ptr()),
                                                      ctypes.c_void_p(inverse.data_ptr()),
                                                      ctypes.c_void_p(self._loss.data_ptr()),
                                                      ctypes.c_void_p(self._gV.data_ptr()), self._stream()))
        return self._loss, self._gV

    def scatter_apply(self, ids, rows):
        L = self._lib.lib()
        n = ids.numel()
        need = int(L.nncf_scatter_apply_ws_ints(n))
        if getattr(self, "_sws", None) is None or self._sws.numel() < need:
            self._sws = torch.empty(need, dtype=torch.int32, device=self.device)
        self._lib.check(L.nncf_scatter_apply(ctypes.c_void_p(self.V.data_ptr()), self.dim, self.dim,
                                             ctypes.c_void_p(ids.data_ptr()), n, ctypes.c_void_p(rows.data_ptr()),
                                             self.lr, ctypes.c_void_p(self._sws.data_ptr()), self._stream()))


class ShardedNegShared(object):
    """Collective neg_shared trainer state of one rank (see module docstring)."""

    def __init__(self, backend, user_count, item_count, batch_size_p, dim, group=None):
        self.be = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.us = shard_size(user_count, self.world)
        self.isz = shard_size(item_count, self.world)
        self.u0 = self.rank * self.us
        self.i0 = self.rank * self.isz
        self.b = int(batch_size_p)
        self.dim = int(dim)
        self.cap = self.b  # ids requested from one peer per step: at most b (all uniques on one shard)

    def local_counts(self, user_count, item_count):
        return (max(0, min(self.us, user_count - self.u0)), max(0, min(self.isz, item_count - self.i0)))

    def _a2a(self, out, inp):
        if self.world == 1:
            out.copy_(inp)
        else:
            dist.all_to_all_single(out, inp, group=self.group)

    def step(self, uid, cid):
        """uid, cid: int32 tensors [b] on the backend's device (global ids; users in my range).
        Returns the local loss tensor [1] (not synchronised)."""
        be, G, cap, d = self.be, self.world, self.cap, self.dim
        dev = uid.device
        uniq, inverse, n_u = be.unique(cid)                       # uniq padded with -1 past n_u
        # -- routing table: slot (owner, k) of every unique id, owners visited in id order --------
        owner = torch.where(uniq >= 0, uniq // self.isz, torch.full_like(uniq, G))
        order = torch.argsort(owner, stable=True)                 # unique indices grouped by owner
        owner_s = owner[order]
        counts = torch.bincount(owner_s.clamp(max=G), minlength=G + 1)[:G]
        starts = torch.cumsum(counts, 0) - counts
        k_in_owner = torch.arange(uniq.numel(), device=dev) - starts[owner_s.clamp(max=G - 1)]
        live = owner_s < G
        flat = (owner_s.clamp(max=G - 1) * cap + k_in_owner)[live].long()   # slot in the [G][cap] request
        src = order[live].long()                                              # unique index held by that slot
        req = torch.full((G * cap,), -1, dtype=torch.int32, device=dev)
        req[flat] = uniq[src]
        # -- all-to-all #1: ids to owners; #2: rows back ----------------------------------------
        req_in = torch.empty_like(req)
        self._a2a(req_in, req)
        local_ids = torch.where(req_in >= 0, req_in - self.i0, req_in)
        rows_out = be.gather_rows(local_ids.contiguous())                     # [G*cap][d]
        rows_in = torch.empty_like(rows_out)
        self._a2a(rows_in, rows_out)
        vrows = torch.zeros(self.b, d, dtype=rows_in.dtype, device=dev)
        vrows[src] = rows_in[flat]                                            # unique order (tf.unique ranks)
        # -- local compute ----------------------------------------------------------------------
        loss, gV = be.local_step((uid - self.u0).to(torch.int32).contiguous(), inverse, vrows)
        # -- all-to-all #3: item gradients to owners, deterministic apply ------------------------
        g_out = torch.zeros(G * cap, d, dtype=gV.dtype, device=dev)
        g_out[flat] = gV[src]
        g_in = torch.empty_like(g_out)
        self._a2a(g_in, g_out)
        be.scatter_apply(local_ids.contiguous(), g_in)
        return loss
