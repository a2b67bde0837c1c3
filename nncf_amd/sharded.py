"""Multi-GPU neg_shared training: one process per GPU, tables range-sharded, RCCL all-to-all
over xGMI for the cross-shard row exchange (no reference counterpart --
the reference is single-process).

Partitioning
  * users: contiguous id ranges; rank r owns users [r*Us, (r+1)*Us) AND every training link of
    those users, so user rows are always local.
  * items: contiguous id ranges; rank r owns item rows [r*Is, (r+1)*Is) (north-star: "the
    embedding table shards by item-id range").
One global step (all ranks call `step` together, weak scaling: b positives per rank)
  1. each rank de-duplicates its batch's items and lays the unique ids out per owner
     (nncf_shard_route: request slot = owner * b + rank among the owner's ids);
  2. all-to-all #1: requested item ids -> owners      (fixed capacity b per peer, -1 padded)
  3. owners gather the rows (HIP), all-to-all #2: rows -> requesters, landing directly in the
     buffer the step kernels read as their "item table" (batch positions index request slots)
  4. local neg_shared step (tile + apply kernels) on [local user shard, fetched item rows];
     user rows are updated in place, item gradients are exported per request slot
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

    def __init__(self, engine_kwargs, local_users, local_items, device, world, item_shard, item_offset, seed=0):
        from . import _lib
        from .engine import Engine
        self._lib = _lib
        self.device = torch.device("cuda", device)
        kw = dict(engine_kwargs)
        if kw.get("interaction_bias"):
            raise NotImplementedError("interaction_bias is not built for the item-sharded path")
        kw.pop("bias_item_count", None)
        kw["user_count"] = local_users
        self.b = kw["batch_size_p"]
        self.dim = (int(kw["dim"]) + 3) // 4 * 4   # physical row width (Engine pads to a multiple of 4)
        self.world, self.cap = int(world), self.b
        self.item_shard, self.i0 = int(item_shard), int(item_offset)
        # the model's "item table" is the landing buffer of fetched rows: world * cap request slots
        kw["item_count"] = self.world * self.cap
        self.engine = Engine(device=device, seed=seed, **kw)
        self.U = self.engine.U                      # local user shard
        self.rows_in = self.engine.V                # [world*cap][d], filled by all-to-all #2
        gen = torch.Generator().manual_seed(int(seed) + 1)
        V = torch.zeros(local_items, self.dim)
        V[:, :int(kw["dim"])] = (torch.rand(local_items, int(kw["dim"]), generator=gen) * 2 - 1) * 0.05
        self.V = V.to(self.device).contiguous()
        self.lr = float(kw["learn_rate"])
        n = self.world * self.cap     # the owner side merges the ranks' sorted request blocks (nncf_scatter_merge): no plan, no cap on n
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=self.device)  # noqa: E731
        self.req = i32(n)
        self.slot = i32(self.b)
        self.rows_out = torch.empty(n, self.dim, dtype=torch.float32, device=self.device)
        self.g_out = torch.zeros(n, self.dim, dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        L = _lib.lib()
        self._ws_route = i32(int(L.nncf_scatter_apply_ws_ints(self.b)))

    def ids(self, x):
        return self.engine.ids(x)

    def _stream(self):
        return self._lib.stream_ptr(torch.cuda.current_stream(self.device))

    def route(self, cid):
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self._lib.check(self._lib.lib().nncf_shard_route(p(cid), cid.numel(), self.item_shard, self.world, self.cap,
                                                         p(self.req), p(self.slot), p(self._ws_route), self._stream()))
        return self.req, self.slot

    def gather_rows(self, req_in):
        """rows of the LOCAL item shard for the requested global ids (zeros where id < 0)."""
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self._lib.check(self._lib.lib().nncf_gather_rows(p(self.V), self.dim, self.dim, p(req_in), req_in.numel(),
                                                         self.i0, p(self.rows_out), self._stream()))
        return self.rows_out

    def rows_landing(self):
        return self.rows_in

    def local_step(self, uid_local, slot):
        """neg_shared step on (local user shard, fetched rows): (loss[1], item grads [world*cap][d])."""
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self._lib.check(self._lib.lib().nncf_step_neg_shared_export(self.engine._h, p(uid_local), p(slot),
                                                                    p(self._loss), p(self.g_out), self._stream()))
        return self._loss, self.g_out

    def scatter_apply(self, req_in, g_in):
        """owner side: req_in [world][cap] as the all-to-all delivered it (every rank's block ascending in id, -1 padded),
        g_in the matching gradient rows: rows of an id summed in rank order, the shard row stepped once"""
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        self._lib.check(self._lib.lib().nncf_scatter_merge(p(self.V), self.dim, self.dim, p(req_in), self.world, self.cap,
                                                           self.i0, p(g_in), self.lr, self._stream()))

    # -- chunked protocol (nncf_shard_*): plans and the id exchange of up to CHUNK steps batched ------------
    CHUNK = 64

    def _session(self):
        if getattr(self, "_sh", None) is None:
            L = self._lib.lib()
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                self._lib.check(L.nncf_shard_create(self.engine._h, ctypes.c_void_p(self.V.data_ptr()), self.dim, self.item_shard,
                                                    self.i0, self.world, self.device.index, ctypes.byref(h)))
            ptrs = [ctypes.c_void_p() for _ in range(6)]
            self._lib.check(L.nncf_shard_buffers(h, *[ctypes.byref(x) for x in ptrs]))
            n = self.world * self.cap

            def view(ptr, count, dtype, shape):
                # a torch view of library-owned device memory (the collectives read / write it in place)
                t = _wrap_device_pointer(ptr.value, count, dtype, self.device)
                return t.view(*shape)
            self._sh = h
            self._bufs = dict(
                req_out=view(ptrs[0], n * self.CHUNK, torch.int32, (self.world, self.CHUNK * self.cap)),
                req_in=view(ptrs[1], n * self.CHUNK, torch.int32, (self.world, self.CHUNK * self.cap)),
                rows_out=view(ptrs[2], n * self.dim, torch.float32, (n, self.dim)),
                rows_in=view(ptrs[3], n * self.dim, torch.float32, (n, self.dim)),
                g_out=view(ptrs[4], n * self.dim, torch.float32, (n, self.dim)),
                g_in=view(ptrs[5], n * self.dim, torch.float32, (n, self.dim)))
        return self._sh

    def chunk_buffers(self):
        self._session()
        return self._bufs

    def plan_chunk(self, uid_chunk, cid_chunk, n_steps):
        h = self._session()
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_plan(h, p(uid_chunk), p(cid_chunk), int(n_steps), self._stream()))

    def plan_owner(self, n_steps):
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_plan_owner(self._session(), int(n_steps), self._stream()))

    def local_step_at(self, index, uid_step, loss_out):
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_step(self._session(), int(index), ctypes.c_void_p(uid_step.data_ptr()),
                                                            ctypes.c_void_p(loss_out.data_ptr()), self._stream()))

    def apply_at(self, index, n_steps):
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_apply(self._session(), int(index), int(n_steps), self.lr, self._stream()))

    def close(self):
        h, self._sh = getattr(self, "_sh", None), None
        if h:
            self._lib.lib().nncf_shard_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _wrap_device_pointer(ptr, count, dtype, device):
    """torch tensor over `count` elements of device memory owned by the native library (no copy, no ownership)."""
    class _Holder(object):
        pass
    holder = _Holder()
    holder.__cuda_array_interface__ = {
        "shape": (int(count),), "typestr": {torch.int32: "<i4", torch.float32: "<f4"}[dtype], "data": (int(ptr), False),
        "version": 3, "strides": None}
    return torch.as_tensor(holder, device=device)


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
        self._req_in = None
        self._g_in = None

    def _a2a(self, out, inp):
        if out.data_ptr() == inp.data_ptr():
            return                      # one rank: the native session aliases the two ends
        if self.world == 1:
            out.copy_(inp)
        else:
            dist.all_to_all_single(out, inp, group=self.group)

    def run_epoch(self, uid_local_all, cid_all, n_steps, losses=None):
        """n_steps global steps over this rank's link stream (ids of n_steps * b links on the backend's device), with
        the plans and the id all-to-all of up to CHUNK steps batched: per step only rows -> local step -> gradients ->
        apply remain.  Backends without the chunk protocol fall back to step()."""
        be, b = self.be, self.b
        if not hasattr(be, "plan_chunk"):
            for t in range(n_steps):
                l = self.step(uid_local_all[t * b:(t + 1) * b], cid_all[t * b:(t + 1) * b])
                if losses is not None:
                    losses[t:t + 1].copy_(l.reshape(1))
            return losses
        bufs = be.chunk_buffers()
        ch = be.CHUNK
        for t0 in range(0, n_steps, ch):
            cnt = min(ch, n_steps - t0)
            be.plan_chunk(uid_local_all[t0 * b:(t0 + cnt) * b], cid_all[t0 * b:(t0 + cnt) * b], cnt)
            self._a2a(bufs["req_in"], bufs["req_out"])                  # all-to-all #1: the chunk's ids -> owners
            be.plan_owner(cnt)                                           # received ids regrouped per step; rows of step 0 gathered
            for i in range(cnt):
                t = t0 + i
                self._a2a(bufs["rows_in"], bufs["rows_out"])            # all-to-all #2: rows -> requesters
                lo = losses[t:t + 1] if losses is not None else be._loss
                be.local_step_at(i, uid_local_all[t * b:(t + 1) * b], lo)
                self._a2a(bufs["g_in"], bufs["g_out"])                  # all-to-all #3: gradients -> owners
                be.apply_at(i, cnt)                                      # apply; rows of step i + 1 gathered
        return losses

    def step(self, uid_local, cid):
        """uid_local: int32 [b] user ids RELATIVE to this rank's user range; cid: int32 [b] global
        item ids; both on the backend's device.  Returns the local loss tensor [1] (no host sync)."""
        be = self.be
        req, slot = be.route(cid)                                  # request slots, slot of every batch position
        if self._req_in is None:
            self._req_in = torch.empty_like(req)
        self._a2a(self._req_in, req)                               # all-to-all #1: ids -> owners
        rows_out = be.gather_rows(self._req_in)
        self._a2a(be.rows_landing(), rows_out)                     # all-to-all #2: rows -> requesters
        loss, g_out = be.local_step(uid_local, slot)               # user rows updated, item grads exported
        if self._g_in is None:
            self._g_in = torch.empty_like(g_out)
        self._a2a(self._g_in, g_out)                               # all-to-all #3: gradients -> owners
        be.scatter_apply(self._req_in, self._g_in)                 # deterministic (rank, slot) order
        return loss


# ------------------------------------------------------------------------------------------
# basic_embedding on several GPUs: data parallel
# ------------------------------------------------------------------------------------------
class HipTowerReplica(object):
    """This rank's replica of a basic_embedding model (user table + text tower) on its GPU,
    stepping through nncf_tower_step with the tower's gradients exported instead of applied."""

    def __init__(self, model_dict, scheme="neg_shared"):
        from . import _lib
        self._lib = _lib
        key = {"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]
        self.head = model_dict[key]
        self.core, self.scheme = self.head.core, scheme
        self.engine, self.tower = self.core.engine, self.core.tower
        L = _lib.lib()
        n = int(L.nncf_tower_grad_floats(self.tower.handle))
        self.grad = torch.zeros(n, dtype=torch.float32, device=self.engine.device)
        _lib.check(L.nncf_tower_set_grad_export(self.tower.handle, ctypes.c_void_p(self.grad.data_ptr())))
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.engine.device)
        self.PLAN_CHUNK = 64 if scheme in self._SCHEME else 0   # steps planned per launch (original: per step)

    _SCHEME = {"neg_shared": 1, "group_neg_shared": 2}

    def step(self, uid, cid, loss_out=None):
        return self.core.step(self.scheme, uid, cid, update=True, loss_out=loss_out if loss_out is not None else self._loss)

    def plan(self, uid_chunk, cid_chunk, n_steps):
        """Duplicate plans + tf.unique of up to 64 consecutive steps in one launch (nncf_tower_plan_steps)."""
        e, L = self.engine, self._lib.lib()
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(e.device):
            self._lib.check(L.nncf_tower_plan_steps(self.tower.handle, e._h, self._SCHEME[self.scheme], p(uid_chunk), p(cid_chunk),
                                                    None, int(n_steps), e._stream()))

    def step_at(self, index, uid, loss_out=None):
        e, L = self.engine, self._lib.lib()
        out = loss_out if loss_out is not None else self._loss
        with torch.cuda.device(e.device):
            self._lib.check(L.nncf_tower_step_at(self.tower.handle, e._h, self._SCHEME[self.scheme], int(index),
                                                 ctypes.c_void_p(uid.data_ptr()), ctypes.c_void_p(out.data_ptr()), e._stream()))
        self.tower.version += 1
        return out

    def apply(self, scale=1.0):
        e = self.engine
        with torch.cuda.device(e.device):
            self._lib.check(self._lib.lib().nncf_tower_apply_grads(self.tower.handle, ctypes.c_void_p(self.grad.data_ptr()),
                                                                   float(scale), 0, e._stream()))


class DataParallelTower(object):
    """Synchronous data-parallel SGD for model_choice = basic_embedding (no reference counterpart).

    The model's large state is the user table; the item side is a text tower whose parameters
    (word table, Dense, BatchNorm: ~3 MB at CiteULike shape) are shared by every item, so item-id
    sharding has nothing to shard: every rank keeps a replica of the tower.  Users and their links
    are partitioned by user range (user rows are local, never exchanged).  One global step:
      1. every rank runs its batch through the whole step (plan, tower forward, U.V^T tile, sparse
         SGD on its user rows, tower backward) with the tower's gradients written to one flat buffer;
      2. ONE all-reduce (sum) of that buffer -- the only data-path collective, a real exchange: the
         parameters are shared;
      3. every rank applies the summed gradient (same arithmetic on every rank: replicas stay
         bit-identical).
    BatchNorm uses each rank's batch statistics (as every data-parallel BN does); running statistics
    stay local.  Weak scaling: b positives per rank."""

    def __init__(self, replica, group=None):
        self.be = replica
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self, uid, cid, loss_out=None):
        loss = self.be.step(uid, cid, loss_out)
        if self.world > 1:
            dist.all_reduce(self.be.grad, op=dist.ReduceOp.SUM, group=self.group)
        self.be.apply(1.0)
        return loss

    def run_epoch(self, uid_all, cid_all, n_steps, b, losses=None):
        """n_steps global steps over this rank's link stream (uid_all / cid_all: n_steps * b ids).  Replicas
        that can plan ahead (HipTowerReplica) make the duplicate plans of 64 steps in one launch, as the
        single-GPU epoch runner does; per step: planned step -> all-reduce -> apply."""
        be = self.be
        chunk = getattr(be, "PLAN_CHUNK", 0)
        for t0 in range(0, n_steps, max(chunk, 1)):
            cnt = min(max(chunk, 1), n_steps - t0)
            if chunk:
                be.plan(uid_all[t0 * b:(t0 + cnt) * b], cid_all[t0 * b:(t0 + cnt) * b], cnt)
            for i in range(cnt):
                t = t0 + i
                lo = losses[t:t + 1] if losses is not None else None
                if chunk:
                    be.step_at(i, uid_all[t * b:(t + 1) * b], lo)
                else:
                    be.step(uid_all[t * b:(t + 1) * b], cid_all[t * b:(t + 1) * b], lo)
                if self.world > 1:
                    dist.all_reduce(be.grad, op=dist.ReduceOp.SUM, group=self.group)
                be.apply(1.0)
