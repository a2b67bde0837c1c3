"""Multi-GPU neg_shared training: one process per GPU, tables range-sharded, RCCL all-to-all
over xGMI for the cross-shard row exchange (no reference counterpart --
the reference is single-process, main.py:34-35).

Partitioning (both forms)
  * users: contiguous id ranges; rank r owns user rows [r*Us, (r+1)*Us)
  * items: contiguous id ranges; rank r owns item rows [r*Is, (r+1)*Is) (north-star: "the
    embedding table shards by item-id range").

WEAK form (`ShardedNegShared`; weak scaling: b positives per rank and step).  Rank r also owns every training
link of its users, so user rows are always local.  One global step (all ranks together):
  1. each rank de-duplicates its batch's items; landing slot of a unique id = its rank among the
     batch's unique ids (ascending); owners are id ranges, so the ids of owner 0, 1, ... are contiguous runs;
  2. requested ids -> owners                      (all-to-all; batched for 64 steps at a time)
  3. owners gather the rows (HIP), rows -> requesters, landing directly in the buffer the step
     kernel reads as its "item table"            (all-to-all)
  4. local neg_shared step on [local user shard, fetched item rows]: user rows are updated in
     place, item gradients are exported per landing slot
  5. item gradients -> owners                     (all-to-all), which add the ranks' rows of an id in rank order and
     apply the update -- sparse SGD, or the model's Adam with the shard's own moments (deterministic, no atomics).
This is synchronous data-parallel SGD: all ranks' batches see the same pre-step item rows and
the owners apply the sum of the per-rank gradients (`simulate_global_steps` in tests/test_sharded_gloo.py restates it).

STRONG form (`ShardedGlobalBatch`; strong scaling: ONE batch of b positives per step over all ranks --
SURVEY.md 8(e) as written, and the only form whose result can be compared with the one-device step on identical
inputs, models/train_neg_shared.py:46-51).  Rank g holds rows g, g + G, ... of the b x b score matrix: it fetches the
user rows of its b / G positions AND the item rows of all b columns from their owners, runs the rectangular form of
the one-launch step (StepArgs::bc), and returns both gradients; the owners add the ranks' partial rows in rank order.
The step's loss is the sum of the ranks' shares.

The wire format is compact: every all-to-all carries exactly the rows that are needed (variable splits; the
per-(peer, step) counts are exchanged once per 64 steps together with the ids, which costs one host read of
2 x world x 64 integers per chunk).  xGMI is a full mesh, so all-to-all drives every link at once, unlike a ring
all-reduce.  The exchange logic is backend-agnostic: `HipBackend` runs the kernels, tests inject a numpy
backend and run the same code over gloo on CPU; with a gloo process group and CUDA tensors the payloads are staged
through host memory (two ranks sharing one GPU in the tests).
"""
import contextlib
import ctypes
import os

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
    """Compute on this rank's GPU through libnncf_hip.so.

    mode = "weak": the engine's user table is this rank's user shard, its item table the landing buffer.
    mode = "strong": both engine tables are landing buffers (b / world user rows, b item rows); the shards are
    `self.U` (local_users rows) and `self.V` (local_items rows).  Pairwise losses (log-loss, max-margin) run the strong form
    REPLICATED (`self.replicated`): the gradient of a column needs sums over all ranks' rows, so every rank fetches the rows
    of ALL b users and b items, runs the whole b x b step on its landing buffers (the one-device kernels) and the owners write
    their rows back -- no gradient exchange, compute redundant world times: the parity form (csrc/rows.hip).  An
    `interaction_bias` takes the same road for every loss (the bias values travel with the rows: `self.bu` / `self.bi` are this
    rank's bias shards, sparse SGD); with an Adam optimiser the moment rows travel with the table rows (`self._adam`)."""

    def __init__(self, engine_kwargs, local_users, local_items, device, world, item_shard, item_offset, seed=0, rank=0,
                 mode="weak", user_shard=None, user_offset=0, shared_device=False):
        from . import _lib
        from .engine import Engine
        assert mode in ("weak", "strong")
        self._lib = _lib
        self.device = torch.device("cuda", device)
        self.mode, self.rank = mode, int(rank)
        kw = dict(engine_kwargs)
        self.bias = kw.get("interaction_bias")          # None | 'user' | 'item' | 'both' (interaction_dot.py:62-71)
        if self.bias and mode != "strong" and kw.get("optimizer", "sgd") != "sgd":
            raise NotImplementedError("interaction_bias on the weak sharded form steps with sparse SGD (the bias moments do not travel)")
        kw.pop("bias_item_count", None)
        self.b = kw["batch_size_p"]
        self.dim = (int(kw["dim"]) + 3) // 4 * 4   # physical row width (Engine pads to a multiple of 4)
        self.world, self.cap = int(world), self.b
        self.item_shard, self.i0 = int(item_shard), int(item_offset)
        self.user_shard, self.u0 = int(user_shard or local_users), int(user_offset)
        self.replicated = mode == "strong" and (kw["loss"] in ("log-loss", "max-margin") or bool(self.bias))
        if self.replicated:
            if kw.get("optimizer", "sgd") != "sgd" and self.bias:
                raise NotImplementedError("interaction_bias on the strong form runs replicated with sparse SGD (the rows' Adam moments "
                                          "travel with them, the bias vectors' do not)")
            kw["user_count"] = self.b                        # landing rows of ALL positions
            kw["item_count"] = self.b
        elif mode == "strong":
            assert self.b % (16 * self.world) == 0, "batch_size_p must be a multiple of 16 * world"
            kw["user_count"] = self.b // self.world          # landing rows of this rank's positions
            kw["item_count"] = self.b                        # landing rows of all columns
        else:
            kw["user_count"] = local_users
            # the model's "item table" is the landing buffer of fetched rows; world * cap slots for the per-step protocol
            # (step(): static capacity b per peer), of which the chunked protocol uses the first b
            kw["item_count"] = self.world * self.cap
        self.optimizer = kw.pop("optimizer", "sgd")     # 'sgd' | 'sparse_adam' | 'adam' (Engine.set_optimizer), chunked protocol only
        self.engine = Engine(device=device, seed=seed, **kw)
        if shared_device:
            self.engine.set_shared_device(True)
        if self.optimizer != "sgd":
            # the engine's own state: the weak form's local user rows (its landing buffers never take a step)
            self.engine.set_optimizer(self.optimizer)
        self.rows_in = self.engine.V                # landing buffer of item rows
        dim_l = int(kw["dim"])

        def table(rows, s):
            gen = torch.Generator().manual_seed(int(s))
            t = torch.zeros(rows, self.dim)
            t[:, :dim_l] = (torch.rand(rows, dim_l, generator=gen) * 2 - 1) * 0.05
            return t.to(self.device).contiguous()
        self.V = table(local_items, int(seed) + 1)   # this rank's item shard
        self.U = table(local_users, int(seed) + 2) if mode == "strong" else self.engine.U   # this rank's user shard
        # this rank's shards of the InteractionDot bias vectors (zero-initialised, interaction_dot.py:62-71); the engine's own
        # vectors are their landing buffers (indexed by landing slot)
        zb = lambda rows: torch.zeros(rows, dtype=torch.float32, device=self.device)  # noqa: E731
        if mode == "strong":
            self.bu = zb(local_users) if self.bias in ("user", "both") else None
        else:   # weak form: the users are local -- the engine's own user-bias vector IS the shard (stepped in place)
            self.bu = self.engine.bias["ub"] if self.bias in ("user", "both") else None
        self.bi = zb(local_items) if self.bias in ("item", "both") else None
        self.lr = float(kw["learn_rate"])
        n = self.world * self.cap     # per-step protocol: the owner side merges the ranks' sorted request blocks (nncf_scatter_merge)
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=self.device)  # noqa: E731
        if mode == "weak":
            self.req = i32(n)
            self.slot = i32(self.b)
            self.rows_out = torch.empty(n, self.dim, dtype=torch.float32, device=self.device)
            self.g_out = torch.zeros(n, self.dim, dtype=torch.float32, device=self.device)
            self._ws_route = i32(int(_lib.lib().nncf_scatter_apply_ws_ints(self.b)))
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    def ids(self, x):
        return self.engine.ids(x)

    def _stream(self):
        return self._lib.stream_ptr(torch.cuda.current_stream(self.device))

    # -- per-step protocol (weak form; static capacity b per peer, -1 padded) ------------------------------
    def route(self, cid):
        if self.optimizer != "sgd":
            raise NotImplementedError("the per-step protocol (static splits) steps the owners' rows with sparse SGD only; "
                                      "run_epoch (the chunked protocol) carries the Adam state of the shards")
        if self.bias:   # the landing bias vector would be stepped in place and never reach its owner: refuse, do not mis-train
            raise NotImplementedError("the per-step protocol (ShardedNegShared.step) has no interaction_bias form; "
                                      "run_epoch (the chunked protocol) moves the item bias values and their gradients")
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

    # -- chunked protocol (nncf_shard_*): plans and the id exchange of up to CHUNK steps batched, compact payloads ------
    CHUNK = 64

    @property
    def nx(self):
        return 2 if self.mode == "strong" else 1

    def _session(self):
        if getattr(self, "_sh", None) is None:
            L = self._lib.lib()
            h = ctypes.c_void_p()
            strong = self.mode == "strong"
            with torch.cuda.device(self.device):
                self._lib.check(L.nncf_shard_create(
                    self.engine._h, ctypes.c_void_p(self.V.data_ptr()), self.dim, self.item_shard, self.i0,
                    ctypes.c_void_p(self.U.data_ptr()) if strong else ctypes.c_void_p(), self.dim, self.user_shard, self.u0,
                    self.world, self.rank, self.device.index, ctypes.byref(h)))
            self._sh = h
            if self.bias:
                pb = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()  # noqa: E731
                self._lib.check(L.nncf_shard_set_bias(h, pb(self.bu) if strong else ctypes.c_void_p(), pb(self.bi)))
            if self.optimizer != "sgd":   # Adam state of this rank's shards (items; users too in the strong form)
                dense = self.optimizer == "adam"
                self._adam = []
                for xi, tab in enumerate([self.V] + ([self.U] if strong else [])):
                    m_, v_ = torch.zeros_like(tab), torch.zeros_like(tab)
                    st_ = torch.zeros(tab.shape[0], dtype=torch.int32, device=self.device) if dense else None
                    self._adam.append((m_, v_, st_))
                    self._lib.check(L.nncf_shard_set_optimizer(h, xi, ctypes.c_void_p(m_.data_ptr()), ctypes.c_void_p(v_.data_ptr()),
                                                               ctypes.c_void_p(st_.data_ptr()) if dense else ctypes.c_void_p()))
            W, CH = self.world, self.CHUNK
            co, ci, nx = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32()
            self._lib.check(L.nncf_shard_counts(h, ctypes.byref(co), ctypes.byref(ci), ctypes.byref(nx)))
            assert nx.value == self.nx

            def view(ptr, count, dtype, shape):
                # a torch view of library-owned device memory (the collectives read / write it in place)
                return _wrap_device_pointer(ptr.value, count, dtype, self.device).view(*shape)
            bufs = dict(cnt_out=view(co, W * self.nx * CH, torch.int32, (W, self.nx * CH)),
                        cnt_in=view(ci, W * self.nx * CH, torch.int32, (W, self.nx * CH)), x=[])
            for xi in range(self.nx):
                ptrs = [ctypes.c_void_p() for _ in range(6)]
                n = ctypes.c_int32()
                self._lib.check(L.nncf_shard_buffers(h, xi, *([ctypes.byref(q) for q in ptrs] + [ctypes.byref(n)])))
                n = n.value
                own = W * n                      # owner-side capacity (W = 1: n, aliased or not)
                bufs["x"].append(dict(
                    n=n,
                    req_out=view(ptrs[0], CH * n, torch.int32, (CH * n,)),
                    req_in=view(ptrs[1], CH * own, torch.int32, (CH * own,)),
                    rows_out=view(ptrs[2], own * self.dim, torch.float32, (own, self.dim)),
                    rows_in=view(ptrs[3], n * self.dim, torch.float32, (n, self.dim)),
                    g_out=view(ptrs[4], n * self.dim, torch.float32, (n, self.dim)),
                    g_in=view(ptrs[5], own * self.dim, torch.float32, (own, self.dim))))
                mb = [ctypes.c_void_p() for _ in range(4)]
                self._lib.check(L.nncf_shard_moment_buffers(h, xi, *[ctypes.byref(q) for q in mb]))
                if mb[0].value:   # Adam on the replicated form: the moment rows of the same ids
                    bufs["x"][-1].update(m_out=view(mb[0], own * self.dim, torch.float32, (own, self.dim)),
                                         m_in=view(mb[1], n * self.dim, torch.float32, (n, self.dim)),
                                         v_out=view(mb[2], own * self.dim, torch.float32, (own, self.dim)),
                                         v_in=view(mb[3], n * self.dim, torch.float32, (n, self.dim)))
                bo, bi_ = ctypes.c_void_p(), ctypes.c_void_p()
                self._lib.check(L.nncf_shard_bias_buffers(h, xi, ctypes.byref(bo), ctypes.byref(bi_)))
                if bo.value:   # the bias values of the same rows (replicated strong form / weak form with an interaction bias)
                    bufs["x"][-1].update(bias_out=view(bo, own, torch.float32, (own,)), bias_in=view(bi_, n, torch.float32, (n,)))
                go, gi = ctypes.c_void_p(), ctypes.c_void_p()
                self._lib.check(L.nncf_shard_bias_grad_buffers(h, xi, ctypes.byref(go), ctypes.byref(gi)))
                if go.value:   # weak form with an item bias: d(loss)/d(cbias) per landing slot travels back with the gradient rows
                    bufs["x"][-1].update(gb_out=view(go, n, torch.float32, (n,)), gb_in=view(gi, own, torch.float32, (own,)))
            self._bufs = bufs
        return self._sh

    def chunk_buffers(self):
        self._session()
        return self._bufs

    def plan_chunk(self, uid_chunk, cid_chunk, n_steps):
        h = self._session()
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_plan(h, p(uid_chunk), p(cid_chunk), int(n_steps), self._stream()))

    def plan_owner(self, n_steps, rows_per_step=None):
        """rows_per_step: int64 [nx][n_steps] host array (rows this shard is asked for per step), None with one rank."""
        ptr = ctypes.c_void_p()
        if rows_per_step is not None:
            self._rps = np.ascontiguousarray(rows_per_step, dtype=np.int64)
            ptr = ctypes.c_void_p(self._rps.ctypes.data)
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_plan_owner(self._session(), int(n_steps), ptr, self._stream()))

    def local_step_at(self, index, uid_step, loss_out):
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_step(self._session(), int(index),
                                                            ctypes.c_void_p(uid_step.data_ptr()) if uid_step is not None else ctypes.c_void_p(),
                                                            ctypes.c_void_p(loss_out.data_ptr()), self._stream()))

    def apply_at(self, index, n_steps):
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_apply(self._session(), int(index), int(n_steps), self.lr, self._stream()))

    def set_comm(self, comm):
        """the transport of run_chunk_native (a Communicator; None: one aliased rank needs none)"""
        self._comm = comm
        self._lib.check(self._lib.lib().nncf_shard_set_comm(self._session(), comm.handle if comm is not None else ctypes.c_void_p()))

    def run_chunk_native(self, uid_chunk, cid_chunk, n_steps, losses):
        """up to CHUNK global steps in ONE native call: plan, the counts / ids exchanges and, per step, rows -> local step ->
        gradients -> apply, every exchange enqueued on the step's stream by the library (nncf_shard_run_chunk)"""
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().nncf_shard_run_chunk(self._session(), p(uid_chunk), p(cid_chunk), int(n_steps),
                                                                 p(losses) if losses is not None else ctypes.c_void_p(), self.lr,
                                                                 self._stream()))

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
        "shape": (int(count),), "typestr": {torch.int32: "<i4", torch.float32: "<f4", torch.uint8: "|u1"}[dtype], "data": (int(ptr), False),
        "version": 3, "strides": None}
    return torch.as_tensor(holder, device=device)


class Communicator(object):
    """nncf_comm_* (include/nncf_hip.h): the native transport of the sharded trainers.
      Communicator.rccl(group, device)      RCCL over xGMI: rank 0 draws the unique id, the process group broadcasts its 128
                                            bytes, every rank joins (one process per GPU; works for ONE rank too)
      Communicator.callback(group, device)  the same native call sequence over a Python callback: device buffers staged through
                                            host memory and exchanged by the process group (gloo) -- the two-ranks-on-one-GPU
                                            tests, where RCCL cannot run (it refuses two ranks on one device)"""

    def __init__(self, handle, kind, keep=None):
        from . import _lib
        self._lib, self.handle, self.kind, self._keep = _lib, handle, kind, keep

    @classmethod
    def rccl(cls, group, device):
        from . import _lib
        L = _lib.lib()
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.nncf_comm_unique_id(buf))
        box = [bytes(buf.raw)]
        if world > 1:
            # broadcast_object_list over nccl stages through the CURRENT device: make it this rank's own
            ctx = torch.cuda.device(torch.device(device)) if torch.device(device).type == "cuda" and torch.device(device).index is not None \
                else contextlib.nullcontext()
            with ctx:
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = ctypes.c_void_p()
        idx = torch.device(device).index
        if idx is None:                      # 'cuda' without an index means the CURRENT device, not device 0
            idx = torch.cuda.current_device()
        _lib.check(L.nncf_comm_create(ctypes.c_char_p(box[0]), world, rank, idx, ctypes.byref(h)))
        return cls(h, "rccl")

    @classmethod
    def callback(cls, group, device):
        from . import _lib
        L = _lib.lib()
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = torch.device(device)

        def xfer(user, d_send, send_cnt, d_recv, recv_cnt, elem_bytes, n_ranks, stream):
            try:
                torch.cuda.synchronize(dev)      # the kernels that produced the payload were enqueued just before
                if elem_bytes == 0:              # all-reduce (sum) of send_cnt[0] floats in place
                    n = int(send_cnt[0])
                    if n:
                        t = _wrap_device_pointer(d_send, n, torch.float32, dev)
                        c = t.cpu()
                        if world > 1:
                            dist.all_reduce(c, group=group)
                        t.copy_(c)
                else:
                    sc = [int(send_cnt[r]) * elem_bytes for r in range(n_ranks)]
                    rc = [int(recv_cnt[r]) * elem_bytes for r in range(n_ranks)]
                    inp = _wrap_device_pointer(d_send, sum(sc), torch.uint8, dev).cpu() if sum(sc) else torch.empty(0, dtype=torch.uint8)
                    out = torch.empty(sum(rc), dtype=torch.uint8)
                    if world > 1:
                        dist.all_to_all_single(out, inp, rc, sc, group=group)
                    else:
                        out.copy_(inp)
                    if sum(rc):
                        _wrap_device_pointer(d_recv, sum(rc), torch.uint8, dev).copy_(out)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:   # noqa: BLE001 -- an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        fn = _lib.XFER_FN(xfer)
        h = ctypes.c_void_p()
        _lib.check(L.nncf_comm_create_callback(fn, ctypes.c_void_p(), world, rank, ctypes.byref(h)))
        return cls(h, "callback", keep=fn)

    def all_reduce_(self, t, stream=None):
        """in-place sum of a float32 device tensor over the ranks, on torch's current stream"""
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._lib.check(self._lib.lib().nncf_comm_all_reduce_f32(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(),
                                                                 self._lib.stream_ptr(stream)))

    def wire_bytes(self, reset=False):
        out = ctypes.c_int64()
        self._lib.check(self._lib.lib().nncf_comm_wire_bytes(self.handle, ctypes.byref(out), 1 if reset else 0))
        return out.value

    def close(self):
        h, self.handle = self.handle, None
        if h:
            self._lib.lib().nncf_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


def _default_transport(native_ok, backend, world):
    """native (RCCL issued by the library) only where it has run on hardware: one nccl rank.  Two or more ranks take torch's
    collectives unless NNCF_SHARD_TRANSPORT=native opts in (the peer-to-peer variable-split exchange between
    distinct GPUs has never executed)."""
    return "native" if (native_ok and backend == "nccl" and world == 1) else "torch"


class _Exchanger(object):
    """all-to-alls of one rank's trainer.  Transports (NNCF_SHARD_TRANSPORT overrides the choice):
      native    the library issues every exchange itself over RCCL, one native call per 64-step chunk (nncf_shard_run_chunk over
                a Communicator.rccl).  Default for ONE nccl rank (the hardware-tested case: NNCF_SHARD_NOALIAS=1); with two or
                more ranks it is OPT-IN (NNCF_SHARD_TRANSPORT=native) until a multi-GPU box has run the two-rank parity tests
                over it -- no such box has been available to the build, and an untested peer-to-peer path must
                not be the one a first multi-GPU run takes silently
      callback  the same native call sequence over Communicator.callback (gloo + host staging): two ranks on one GPU
      torch     this module issues torch.distributed collectives between the library's calls: the process group's own
                transport for device tensors (nccl), host staging for a gloo group; the numpy backends of the CPU tests
    One rank: the native session aliases both ends of every exchange and nothing is exchanged -- unless NNCF_SHARD_NOALIAS=1
    keeps them apart, in which case that ONE rank drives the real collectives (single-GPU tests of the multi-GPU code)."""

    def _init_group(self, group):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self.stage_host = backend == "gloo"
        self._wire = 0            # payload bytes this rank handed to all_to_all_single (incl. its own share), for the record
        native_ok = hasattr(self.be, "run_chunk_native")
        want = os.environ.get("NNCF_SHARD_TRANSPORT", "")
        if want not in ("", "native", "callback", "torch"):
            raise ValueError("NNCF_SHARD_TRANSPORT must be native, callback or torch")
        self.transport = want or _default_transport(native_ok, backend, self.world)
        if self.transport != "torch" and not native_ok:
            raise ValueError("NNCF_SHARD_TRANSPORT=%s needs the HIP backend" % self.transport)
        if self.transport == "native" and backend != "nccl":
            raise ValueError("NNCF_SHARD_TRANSPORT=native needs an nccl process group (got %r)" % (backend,))
        self.comm = None

    @property
    def wire_bytes(self):
        return self._wire + (self.comm.wire_bytes() if self.comm is not None else 0)

    @wire_bytes.setter
    def wire_bytes(self, v):
        self._wire = int(v)
        if self.comm is not None:
            self.comm.wire_bytes(reset=True)

    def _aliased(self):
        bufs = self.be.chunk_buffers()
        return bufs["cnt_in"].data_ptr() == bufs["cnt_out"].data_ptr()

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        if out.data_ptr() == inp.data_ptr():
            return                      # one rank: the native session aliases the two ends
        if out_splits is not None:
            out, inp = out[:int(sum(out_splits))], inp[:int(sum(in_splits))]
            out_splits, in_splits = [int(v) for v in out_splits], [int(v) for v in in_splits]
        if not dist.is_initialized():
            out.copy_(inp)
            return
        self._wire += inp.numel() * inp.element_size()
        if self.stage_host and inp.is_cuda:
            oc = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(oc, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(oc)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def _run_chunks(self, uid_all, cid_all, n_steps, nu, nc, losses, uid_per_step):
        """n_steps global steps; uid_all / cid_all hold nu / nc ids per step.  Per chunk of up to CHUNK steps: plan ->
        counts -> ids; per step: rows -> local step -> gradients -> apply."""
        be = self.be
        bufs = be.chunk_buffers()
        ch, nx, W = be.CHUNK, be.nx, self.world
        exchange = not self._aliased()
        if self.transport in ("native", "callback"):
            if exchange and self.comm is None:
                self.comm = (Communicator.rccl if self.transport == "native" else Communicator.callback)(self.group, be.device)
                be.set_comm(self.comm)
            for t0 in range(0, n_steps, ch):
                cnt = min(ch, n_steps - t0)
                be.run_chunk_native(uid_all[t0 * nu:(t0 + cnt) * nu], cid_all[t0 * nc:(t0 + cnt) * nc], cnt,
                                    losses[t0:t0 + cnt] if losses is not None else None)
            return losses
        for t0 in range(0, n_steps, ch):
            cnt = min(ch, n_steps - t0)
            be.plan_chunk(uid_all[t0 * nu:(t0 + cnt) * nu], cid_all[t0 * nc:(t0 + cnt) * nc], cnt)
            send = recv = None
            if exchange:
                self._a2a(bufs["cnt_in"], bufs["cnt_out"])              # static: nx * CHUNK counts per peer
                both = torch.stack([bufs["cnt_out"], bufs["cnt_in"]]).cpu().numpy().reshape(2, W, nx, ch)   # the chunk's one host read
                send, recv = both[0], both[1]                            # [peer][exchange][step]
                for xi in range(nx):                                     # the chunk's requested ids -> owners
                    xb = bufs["x"][xi]
                    self._a2a(xb["req_in"], xb["req_out"], recv[:, xi, :cnt].sum(1), send[:, xi, :cnt].sum(1))
                be.plan_owner(cnt, recv[:, :, :cnt].sum(0))
            else:
                be.plan_owner(cnt)                                       # rows of step 0 gathered
            for i in range(cnt):
                t = t0 + i
                for xi in range(nx):                                     # rows -> requesters
                    xb = bufs["x"][xi]
                    if exchange:
                        self._a2a(xb["rows_in"], xb["rows_out"], send[:, xi, i], recv[:, xi, i])
                        for a_, b_ in (("m_in", "m_out"), ("v_in", "v_out"), ("bias_in", "bias_out")):
                            if a_ in xb:                                 # ... their Adam moments / bias values
                                self._a2a(xb[a_], xb[b_], send[:, xi, i], recv[:, xi, i])
                    else:
                        self._a2a(xb["rows_in"], xb["rows_out"])
                        for a_, b_ in (("m_in", "m_out"), ("v_in", "v_out"), ("bias_in", "bias_out")):
                            if a_ in xb:
                                self._a2a(xb[a_], xb[b_])
                lo = losses[t:t + 1] if losses is not None else be._loss
                be.local_step_at(i, uid_all[t * nu:(t + 1) * nu] if uid_per_step else None, lo)
                for xi in range(nx if not getattr(be, "replicated", False) else 0):   # gradients -> owners (replicated form: none)
                    xb = bufs["x"][xi]
                    if exchange:
                        self._a2a(xb["g_in"], xb["g_out"], recv[:, xi, i], send[:, xi, i])
                        if "gb_in" in xb:                                # ... and the item-bias gradients of the same slots
                            self._a2a(xb["gb_in"], xb["gb_out"], recv[:, xi, i], send[:, xi, i])
                    else:
                        self._a2a(xb["g_in"], xb["g_out"])
                        if "gb_in" in xb:
                            self._a2a(xb["gb_in"], xb["gb_out"])
                be.apply_at(i, cnt)                                      # apply; rows of step i + 1 gathered
        return losses


class ShardedNegShared(_Exchanger):
    """Collective neg_shared trainer state of one rank, weak form (see module docstring)."""

    def __init__(self, backend, user_count, item_count, batch_size_p, dim, group=None):
        self.be = backend
        self._init_group(group)
        self.us = shard_size(user_count, self.world)
        self.isz = shard_size(item_count, self.world)
        self.u0 = self.rank * self.us
        self.i0 = self.rank * self.isz
        self.b = int(batch_size_p)
        self.dim = int(dim)
        self.cap = self.b  # per-step protocol: ids requested from one peer per step: at most b (all uniques on one shard)
        self._req_in = None
        self._g_in = None

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
        return self._run_chunks(uid_local_all, cid_all, n_steps, b, b, losses, True)

    def step(self, uid_local, cid):
        """uid_local: int32 [b] user ids RELATIVE to this rank's user range; cid: int32 [b] global
        item ids; both on the backend's device.  Returns the local loss tensor [1] (no host sync).
        Per-step protocol with static splits (capacity b per peer, -1 padded)."""
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


class ShardedGlobalBatch(_Exchanger):
    """Strong form: ONE global batch of b links per step, rank g holds positions g, g + G, ... (module docstring).
    Every rank is handed the same global link stream."""

    def __init__(self, backend, user_count, item_count, batch_size_p, dim, group=None):
        self.be = backend
        self._init_group(group)
        self.us = shard_size(user_count, self.world)
        self.isz = shard_size(item_count, self.world)
        self.b = int(batch_size_p)
        self.dim = int(dim)
        self.replicated = bool(getattr(backend, "replicated", False))   # pairwise losses: every rank runs the whole step
        assert self.replicated or self.b % self.world == 0
        self.bl = self.b if self.replicated else self.b // self.world

    def local_positions(self, uid_global_all, n_steps):
        """uid of this rank's positions (g, g + G, ...) of every step: [n_steps * b / G]; replicated form: all positions"""
        if self.replicated:
            return uid_global_all[:n_steps * self.b]
        return uid_global_all[:n_steps * self.b].view(n_steps, self.b)[:, self.rank::self.world].contiguous().view(-1)

    def run_epoch(self, uid_global_all, cid_global_all, n_steps, losses=None):
        """uid / cid: the GLOBAL link stream (n_steps * b ids each, the same on every rank).  losses[t] receives this
        rank's share of step t's loss; the step's loss is the sum over ranks (all-reduce it when you need it)."""
        uid_mine = self.local_positions(uid_global_all, n_steps)
        return self._run_chunks(uid_mine, cid_global_all, n_steps, self.bl, self.b, losses, False)


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

    def run_epoch_native(self, uid_all, cid_all, n_steps, losses, comm=None, scale=1.0):
        """nncf_tower_epoch_dp: n_steps planned steps, each followed by the all-reduce of the exported gradient (issued by the
        library over `comm`, a Communicator; None: one rank) and the optimiser step -- ONE native call for the whole loop."""
        e, L = self.engine, self._lib.lib()
        if losses is None:
            losses = torch.empty(int(n_steps), dtype=torch.float32, device=e.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(e.device):
            self._lib.check(L.nncf_tower_epoch_dp(self.tower.handle, e._h, self._SCHEME[self.scheme], comm.handle if comm is not None else None,
                                                  p(uid_all), p(cid_all), None, int(n_steps), p(losses), float(scale), e._stream()))
        self.tower.version += int(n_steps)
        return losses


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
        backend = dist.get_backend(group) if dist.is_initialized() else None
        hip = hasattr(replica, "tower")            # HipTowerReplica (the numpy replicas of the CPU tests have no device)
        want = os.environ.get("NNCF_SHARD_TRANSPORT", "")
        if want not in ("", "native", "callback", "torch"):
            raise ValueError("NNCF_SHARD_TRANSPORT must be native, callback or torch")
        self.transport = want or _default_transport(hip, backend, self.world)
        if self.transport != "torch" and not hip:
            raise ValueError("NNCF_SHARD_TRANSPORT=%s needs the HIP backend" % self.transport)
        if self.transport == "native" and backend != "nccl":
            raise ValueError("NNCF_SHARD_TRANSPORT=native needs an nccl process group (got %r)" % (backend,))
        # one rank has nothing to reduce -- unless NNCF_SHARD_NOALIAS=1 asks that ONE rank to drive the real collective
        self.reduce = self.world > 1 or (dist.is_initialized() and os.environ.get("NNCF_SHARD_NOALIAS", "0") not in ("", "0"))
        self.comm = None
        if self.reduce and self.transport in ("native", "callback") and hip:
            self.comm = (Communicator.rccl if self.transport == "native" else Communicator.callback)(group, replica.engine.device)
        self.wire_bytes = 0

    def _all_reduce(self):
        g = self.be.grad
        self.wire_bytes += g.numel() * g.element_size()
        if self.comm is not None:
            self.comm.all_reduce_(g)        # RCCL (or the callback transport), enqueued by the library on the step's stream
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)

    def step(self, uid, cid, loss_out=None):
        loss = self.be.step(uid, cid, loss_out)
        if self.reduce:
            self._all_reduce()
        self.be.apply(1.0)
        return loss

    def run_epoch(self, uid_all, cid_all, n_steps, b, losses=None):
        """n_steps global steps over this rank's link stream (uid_all / cid_all: n_steps * b ids).  Replicas
        that can plan ahead (HipTowerReplica) make the duplicate plans of 64 steps in one launch, as the
        single-GPU epoch runner does; per step: planned step -> all-reduce -> apply."""
        be = self.be
        chunk = getattr(be, "PLAN_CHUNK", 0)
        # HIP replicas that can plan ahead run the whole loop natively (round 5): the library issues the all-reduce itself over
        # a native communicator -- or reduces nothing on one aliased rank.  The torch transport keeps the loop below.
        native = chunk and hasattr(be, "run_epoch_native") and (not self.reduce or self.comm is not None) \
            and os.environ.get("NNCF_DP_NATIVE", "1") != "0"
        if native and n_steps > 0:
            # the native runner strides the id streams by the ENGINE's batch size: a different `b`, or streams shorter than
            # n_steps batches, would be a device out-of-bounds read with nothing to report it
            eb = int(be.engine.b)
            if int(b) != eb:
                raise ValueError("run_epoch: b = %d, but the replica's engine was built for batch_size_p = %d" % (b, eb))
            if uid_all.numel() < n_steps * b or cid_all.numel() < n_steps * b:
                raise ValueError("run_epoch: %d steps of %d links need %d ids, got %d / %d" % (n_steps, b, n_steps * b, uid_all.numel(),
                                                                                             cid_all.numel()))
            if losses is not None and losses.numel() < n_steps:
                raise ValueError("run_epoch: losses holds %d entries for %d steps" % (losses.numel(), n_steps))
            if self.reduce:
                self.wire_bytes += int(n_steps) * be.grad.numel() * be.grad.element_size()
            be.run_epoch_native(uid_all[:n_steps * b], cid_all[:n_steps * b], n_steps, losses, comm=self.comm if self.reduce else None)
            return losses      # the caller's tensor (None when none was passed): the same contract as the loop below
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
                if self.reduce:
                    self._all_reduce()
                be.apply(1.0)
        return losses
