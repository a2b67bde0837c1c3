"""CPU, world_size = 2 over gloo: the exchange logic of nncf_amd/sharded.py (routing tables,
three all-to-alls, owner-side deterministic apply) with a numpy compute backend built on the
oracle, against a single-process restatement of one synchronous global step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import step_oracle as so
from nncf_amd.sharded import ShardedNegShared, partition_links_by_user, shard_size

B, D, NU, NI = 32, 8, 40, 60
STEPS = 3


def _conf(loss, bias=None, optimizer="sgd"):
    return so.OracleConf(batch_size_p=B, num_negatives=0, loss=loss, loss_gamma=10. if loss == "log-loss" else None,
                         u_reg=1e-3, learn_rate=0.1 if optimizer == "sgd" else 0.01, model_choice="mf", interaction_bias=bias,
                         optimizer=optimizer)


def _tables(seed=0):
    rs = np.random.RandomState(seed)
    return (rs.uniform(-.5, .5, (NU, D)).astype(np.float32), rs.uniform(-.5, .5, (NI, D)).astype(np.float32))


def _batches(world, seed=1, skew=False):
    """per rank: STEPS batches of B links whose users lie in the rank's range.
    skew: the variable-split cases a one-rank run cannot produce -- rank 0 asks ONLY its own item shard (every other peer gets
    an EMPTY request from it, every step), the last rank asks nine items of shard 0 for every one of its own (unequal splits),
    and in step 1 one single item B times (a one-row exchange); ranks in between ask everybody uniformly."""
    rs = np.random.RandomState(seed)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    out = []
    for r in range(world):
        lo, hi = r * us, min(NU, (r + 1) * us)
        steps = []
        for t in range(STEPS):
            cid = rs.randint(0, NI, B)
            if skew and r == 0:
                cid = rs.randint(0, isz, B)
            elif skew and r == world - 1:
                own = rs.randint(r * isz, NI, B)
                cid = np.where(rs.rand(B) < 0.9, rs.randint(0, isz, B), own)
                if t == 1:
                    cid = np.full(B, 3)
            steps.append((rs.randint(lo, hi, B), cid))
        out.append(steps)
    return out


class NumpyBackend(object):
    """Test double of HipBackend: same interface, oracle arithmetic, CPU tensors."""

    def __init__(self, conf, Ushard, Vshard, world, item_shard, item_offset):
        self.conf, self.U, self.V = conf, Ushard, Vshard
        self.lr = np.float32(conf.learn_rate)
        self.world, self.cap, self.isz, self.i0 = world, conf.batch_size_p, item_shard, item_offset
        self.rows_in = torch.zeros(world * self.cap, Vshard.shape[1])

    def route(self, cid):
        c = cid.numpy()
        req = np.full(self.world * self.cap, -1, dtype=np.int32)
        slot_of = {}
        for o in range(self.world):
            mine = np.unique(c[(c // self.isz) == o])          # ascending ids of owner o
            for k, it in enumerate(mine):
                req[o * self.cap + k] = it
                slot_of[it] = o * self.cap + k
        return torch.from_numpy(req), torch.from_numpy(np.array([slot_of[x] for x in c], dtype=np.int32))

    def gather_rows(self, req_in):
        i = req_in.numpy()
        out = np.zeros((len(i), self.V.shape[1]), dtype=np.float32)
        out[i >= 0] = self.V[i[i >= 0] - self.i0]
        return torch.from_numpy(out)

    def rows_landing(self):
        return self.rows_in

    def local_step(self, uid_local, slot):
        o = so.StepOracle(self.U, self.rows_in.numpy(), self.conf, dtype=np.float32)
        loss = o.step_neg_shared(uid_local.numpy(), slot.numpy())
        self.U[...] = o.U                       # user rows are local: updated in place
        return torch.tensor([loss]), torch.from_numpy(o.last["dV"].astype(np.float32))

    def scatter_apply(self, req_in, rows):
        i, r = req_in.numpy(), rows.numpy()
        acc = {}
        for k in range(len(i)):                 # list order = (rank, slot): deterministic
            if i[k] >= 0:
                acc[i[k]] = acc.get(i[k], np.zeros(r.shape[1], dtype=np.float32)) + r[k]
        for k, g in acc.items():
            self.V[k - self.i0] -= self.lr * g


class _CompactChunks(object):
    """The compact chunk protocol of HipBackend (nncf_shard_*) in numpy: per (peer, step) counts, request lists per owner
    and step holding exactly the unique ids that live there, rows / gradients packed requester-major.  `_tables` lists the
    exchanged tables: (shard array, shard rows per rank, first id, ids per step)."""
    CHUNK = 2     # STEPS = 3: one full chunk and a partial one

    def chunk_buffers(self):
        if not hasattr(self, "_bufs"):
            W, CH, nx = self.world, self.CHUNK, self.nx
            x = []
            for (shard, rows_per_rank, i0, n) in self._tables():
                d = shard.shape[1]
                x.append(dict(n=n, req_out=torch.zeros(CH * n, dtype=torch.int32), req_in=torch.zeros(CH * W * n, dtype=torch.int32),
                              rows_out=torch.zeros(W * n, d), rows_in=torch.zeros(n, d), g_out=torch.zeros(n, d),
                              g_in=torch.zeros(W * n, d)))
            for xb, ms in zip(x, getattr(self, "_moment_shards", lambda: [None] * nx)()):
                if ms is not None:       # Adam on the replicated form: the moment rows of the same ids (nncf_shard_moment_buffers)
                    dd = xb["rows_in"].shape[1]
                    xb.update(m_out=torch.zeros(W * xb["n"], dd), m_in=torch.zeros(xb["n"], dd), v_out=torch.zeros(W * xb["n"], dd),
                              v_in=torch.zeros(xb["n"], dd))
            for xb, bs in zip(x, getattr(self, "_bias_shards", lambda: [None] * nx)()):
                if bs is not None:       # the bias values of the same rows (HipBackend: nncf_shard_bias_buffers)
                    xb.update(bias_out=torch.zeros(W * xb["n"]), bias_in=torch.zeros(xb["n"]))
            self._bufs = dict(cnt_out=torch.zeros(W, nx * CH, dtype=torch.int32), cnt_in=torch.zeros(W, nx * CH, dtype=torch.int32), x=x)
            self._loss = torch.zeros(1)
        return self._bufs

    def _plan_lists(self, lists, n_steps):
        """lists[xi]: flat ids of the chunk, n ids per step"""
        bufs = self.chunk_buffers()
        W, CH = self.world, self.CHUNK
        bufs["cnt_out"].zero_()
        self._slots, self._nuniq = [], []
        for xi, (shard, rpr, i0, n) in enumerate(self._tables()):
            ids = lists[xi].numpy().reshape(n_steps, n)
            per = [[None] * n_steps for _ in range(W)]
            slots, nun = [], []
            for s in range(n_steps):
                uq = np.unique(ids[s])
                slots.append(np.searchsorted(uq, ids[s]).astype(np.int32))
                nun.append(len(uq))
                own = np.minimum(uq // rpr, W - 1)
                for o in range(W):
                    per[o][s] = uq[own == o]
                    bufs["cnt_out"][o, xi * CH + s] = len(per[o][s])
            flat = np.concatenate([per[o][s] for o in range(W) for s in range(n_steps)]).astype(np.int32)
            bufs["x"][xi]["req_out"][:len(flat)] = torch.from_numpy(flat)
            self._slots.append(slots)
            self._nuniq.append(nun)

    def _blocks(self, xi, s):
        """ids this shard is asked for at step s, per requester (from cnt_in / req_in)"""
        bufs, W, CH = self._bufs, self.world, self.CHUNK
        cnt = bufs["cnt_in"].numpy().reshape(W, self.nx, CH)[:, xi, :]
        off = np.concatenate([[0], np.cumsum(cnt.reshape(-1))])
        req = bufs["x"][xi]["req_in"].numpy()
        return [req[off[r * CH + s]:off[r * CH + s] + cnt[r, s]] for r in range(W)]

    def _gather(self, s):
        for xi, (shard, rpr, i0, n) in enumerate(self._tables()):
            ids = np.concatenate(self._blocks(xi, s))
            self._bufs["x"][xi]["rows_out"][:len(ids)] = torch.from_numpy(shard[ids - i0])
            bs = getattr(self, "_bias_shards", lambda: [None] * self.nx)()[xi]
            if bs is not None:
                self._bufs["x"][xi]["bias_out"][:len(ids)] = torch.from_numpy(bs[ids - i0])
            ms = getattr(self, "_moment_shards", lambda: [None] * self.nx)()[xi]
            if ms is not None:
                self._bufs["x"][xi]["m_out"][:len(ids)] = torch.from_numpy(ms[0][ids - i0])
                self._bufs["x"][xi]["v_out"][:len(ids)] = torch.from_numpy(ms[1][ids - i0])

    def plan_owner(self, n_steps, rows_per_step=None):
        if self.world == 1:     # one rank: the native session aliases the two ends; the test double copies
            self._bufs["cnt_in"].copy_(self._bufs["cnt_out"])
            for xb in self._bufs["x"]:
                xb["req_in"][:xb["req_out"].numel()] = xb["req_out"]
        elif rows_per_step is not None:
            for xi in range(self.nx):
                for s in range(n_steps):
                    assert rows_per_step[xi][s] == sum(len(blk) for blk in self._blocks(xi, s))
        self._gather(0)

    def apply_at(self, index, n_steps):
        for xi, (shard, rpr, i0, n) in enumerate(self._tables()):
            g = self._bufs["x"][xi]["g_in"].numpy()
            acc, k = {}, 0
            for blk in self._blocks(xi, index):          # requester order: deterministic
                for it in blk:
                    acc[it] = acc.get(it, np.zeros(g.shape[1], dtype=np.float32)) + g[k]
                    k += 1
            for it, gr in acc.items():
                shard[it - i0] -= self.lr * gr
        if index + 1 < n_steps:
            self._gather(index + 1)


class ChunkedNumpyBackend(_CompactChunks, NumpyBackend):
    """weak form: items exchanged, user rows local.  With an InteractionDot bias (conf.interaction_bias; round 6): the user bias
    vector is local like the user rows, the item bias values travel with the item rows and their gradients back with the
    gradient rows (HipBackend: nncf_shard_set_bias / nncf_shard_bias_grad_buffers on the weak form)."""
    nx = 1

    def _tables(self):
        return [(self.V, self.isz, self.i0, self.cap)]

    def set_bias(self, ub_local, cb_shard):
        self.ub, self.cb = ub_local, cb_shard      # float32 arrays or None

    def _bias_shards(self):
        return [getattr(self, "cb", None)]

    def chunk_buffers(self):
        first = not hasattr(self, "_bufs")
        bufs = _CompactChunks.chunk_buffers(self)
        if first and getattr(self, "cb", None) is not None:
            xb = bufs["x"][0]
            xb.update(gb_out=torch.zeros(xb["n"]), gb_in=torch.zeros(self.world * xb["n"]))
        return bufs

    def plan_chunk(self, uid_chunk, cid_chunk, n_steps):
        self._plan_lists([cid_chunk], n_steps)

    def local_step_at(self, index, uid_step, loss_out):
        xb = self._bufs["x"][0]
        if self.world == 1:
            xb["rows_in"].copy_(xb["rows_out"][:self.cap])
        if self.world == 1 and "bias_in" in xb:
            xb["bias_in"].copy_(xb["bias_out"][:self.cap])
        o = so.StepOracle(self.U, xb["rows_in"].numpy(), self.conf, dtype=np.float32)
        if getattr(self, "ub", None) is not None:
            o.ub = self.ub                                  # local users: stepped in place
        if "bias_in" in xb:
            o.cb = xb["bias_in"].numpy().copy()             # landed values, indexed by landing slot
        loss = o.step_neg_shared(uid_step.numpy(), self._slots[0][index])
        self.U[...] = o.U
        xb["g_out"].copy_(torch.from_numpy(o.last["dV"].astype(np.float32)))
        if "gb_out" in xb:                                  # d(loss)/d(cbias) per landing slot leaves; the owners step
            xb["gb_out"].copy_(torch.from_numpy(o.last_gb["cb"].astype(np.float32)[:self.cap]))
        if self.world == 1:
            xb["g_in"][:self.cap] = xb["g_out"]
            if "gb_out" in xb:
                xb["gb_in"][:self.cap] = xb["gb_out"]
        loss_out.copy_(torch.tensor([loss]).reshape(loss_out.shape))

    def apply_at(self, index, n_steps):
        if getattr(self, "cb", None) is not None:           # the item bias of the requested ids: requester order, then one SGD step
            gb = self._bufs["x"][0]["gb_in"].numpy()
            acc, k = {}, 0
            for blk in self._blocks(0, index):
                for it in blk:
                    acc[it] = acc.get(it, np.float32(0)) + gb[k]
                    k += 1
            for it, g in acc.items():
                self.cb[it - self.i0] -= self.lr * g
        _CompactChunks.apply_at(self, index, n_steps)


class StrongNumpyBackend(_CompactChunks):
    """strong form (one global batch, rank g holds rows g, g + G, ...): users AND items exchanged; the local share
    is oracle/step_oracle.py: neg_shared_row_block"""
    nx = 2

    def __init__(self, conf, Ushard, Vshard, world, rank, user_shard, item_shard):
        self.conf, self.U, self.V, self.world, self.rank = conf, Ushard, Vshard, world, rank
        self.lr = np.float32(conf.learn_rate)
        self.b, self.usz, self.isz = conf.batch_size_p, user_shard, item_shard
        self.bl = self.b // world

    def _tables(self):
        return [(self.V, self.isz, self.rank * self.isz, self.b), (self.U, self.usz, self.rank * self.usz, self.bl)]

    def plan_chunk(self, uid_chunk, cid_chunk, n_steps):
        self._plan_lists([cid_chunk, uid_chunk], n_steps)

    def local_step_at(self, index, uid_step, loss_out):
        xv, xu = self._bufs["x"]
        if self.world == 1:
            xv["rows_in"].copy_(xv["rows_out"][:self.b])
            xu["rows_in"].copy_(xu["rows_out"][:self.bl])
        sv, su = self._slots[0][index], self._slots[1][index]
        rows = self.rank + self.world * np.arange(self.bl)
        loss, g_u, g_v = so.neg_shared_row_block(xu["rows_in"].numpy()[su], xv["rows_in"].numpy()[sv], rows, self.conf)
        gu = np.zeros((self.bl, g_u.shape[1]), dtype=np.float32)
        gv = np.zeros((self.b, g_v.shape[1]), dtype=np.float32)
        np.add.at(gu, su, g_u)                      # per landing slot: positions of one id summed
        np.add.at(gv, sv, g_v)
        xu["g_out"].copy_(torch.from_numpy(gu))
        xv["g_out"].copy_(torch.from_numpy(gv))
        if self.world == 1:
            xu["g_in"][:self.bl] = xu["g_out"]
            xv["g_in"][:self.b] = xv["g_out"]
        loss_out.copy_(torch.tensor([loss]).reshape(loss_out.shape))


class ReplicatedNumpyBackend(_CompactChunks):
    """strong form of the PAIRWISE losses (HipBackend.replicated): every rank fetches the rows of all b users and all b items,
    runs the whole b x b step on them and writes the rows it owns back; no gradient exchange"""
    nx = 2
    replicated = True

    def __init__(self, conf, Ushard, Vshard, world, rank, user_shard, item_shard):
        self.conf, self.U, self.V, self.world, self.rank = conf, Ushard, Vshard, world, rank
        self.lr = np.float32(conf.learn_rate)
        self.b, self.usz, self.isz = conf.batch_size_p, user_shard, item_shard
        ib = getattr(conf, "interaction_bias", None)     # this rank's shards of the InteractionDot bias vectors
        self.bu = np.zeros(Ushard.shape[0], dtype=np.float32) if ib in ("user", "both") else None
        self.bi = np.zeros(Vshard.shape[0], dtype=np.float32) if ib in ("item", "both") else None

    def _tables(self):
        return [(self.V, self.isz, self.rank * self.isz, self.b), (self.U, self.usz, self.rank * self.usz, self.b)]

    def _bias_shards(self):
        return [self.bi, self.bu]

    def _moment_shards(self):
        if getattr(self.conf, "optimizer", "sgd") == "sgd":
            return [None, None]
        if not hasattr(self, "_mom"):     # Adam state of this rank's shards: (m, v) per table; steps taken so far
            self._mom = [(np.zeros_like(self.V), np.zeros_like(self.V)), (np.zeros_like(self.U), np.zeros_like(self.U))]
            self._t = 0
        return self._mom

    def plan_chunk(self, uid_chunk, cid_chunk, n_steps):
        self._ids = [cid_chunk.numpy().reshape(n_steps, self.b), uid_chunk.numpy().reshape(n_steps, self.b)]
        self._plan_lists([cid_chunk, uid_chunk], n_steps)

    def local_step_at(self, index, uid_step, loss_out):
        xv, xu = self._bufs["x"]
        if self.world == 1:
            xv["rows_in"].copy_(xv["rows_out"][:self.b])
            xu["rows_in"].copy_(xu["rows_out"][:self.b])
        sv, su = self._slots[0][index], self._slots[1][index]
        o = so.StepOracle(xu["rows_in"].numpy().copy(), xv["rows_in"].numpy().copy(), self.conf, dtype=np.float32)
        for xb, name in ((xv, "cb"), (xu, "ub")):            # the landed bias values, indexed by landing slot like the rows
            if "bias_in" in xb:
                if self.world == 1:
                    xb["bias_in"].copy_(xb["bias_out"][:self.b])
                setattr(o, name, xb["bias_in"].numpy().copy())
        adam = getattr(self.conf, "optimizer", "sgd") != "sgd"
        if adam:                                             # the landed moment rows are the step's Adam state
            self._moment_shards()
            o.slot_v, o.slot_u = so.AdamSlot(o.V), so.AdamSlot(o.U)
            for xb, slot in ((xv, o.slot_v), (xu, o.slot_u)):
                if self.world == 1:
                    xb["m_in"].copy_(xb["m_out"][:self.b])
                    xb["v_in"].copy_(xb["v_out"][:self.b])
                slot.m[...] = xb["m_in"].numpy()
                slot.v[...] = xb["v_in"].numpy()
            o.t = self._t
            if self.conf.optimizer == "adam":                # Keras' dense Adam: every row of the (landing) table would step; only
                import copy
                o.conf = copy.copy(self.conf)                # the touched ones are real rows here
                o.conf.optimizer = "sparse_adam"
        loss = o.step_neg_shared(su, sv)                     # the one-device step on the landing rows
        for k, ((shard, rpr, i0, n), ids, slots, land, slot) in enumerate(((self._tables()[0], self._ids[0][index], sv, o.V, getattr(o, "slot_v", None)),
                                                                          (self._tables()[1], self._ids[1][index], su, o.U, getattr(o, "slot_u", None)))):
            own = (ids >= i0) & (ids < i0 + shard.shape[0])
            shard[ids[own] - i0] = land[slots[own]]         # the owners write their rows back
            if adam:
                ms, vs = self._mom[k]
                ms[ids[own] - i0] = slot.m[slots[own]]      # ... and the moment rows
                vs[ids[own] - i0] = slot.v[slots[own]]
                if self.conf.optimizer == "adam":            # dense Adam: the rest of the shard takes the zero-gradient step
                    idle = np.setdiff1d(np.arange(shard.shape[0]), ids[own] - i0)
                    lr_t = np.float32(so.adam_lr_t(self.conf.learn_rate, self.conf.beta1, self.conf.beta2, self._t + 1))
                    ms[idle] *= np.float32(self.conf.beta1)
                    vs[idle] *= np.float32(self.conf.beta2)
                    shard[idle] -= lr_t * ms[idle] / (np.sqrt(vs[idle]) + np.float32(self.conf.epsilon))
        if adam:
            self._t += 1
        for bs, (shard, rpr, i0, n), ids, slots, land in ((self.bi, self._tables()[0], self._ids[0][index], sv, o.cb),
                                                          (self.bu, self._tables()[1], self._ids[1][index], su, o.ub)):
            if bs is not None:
                own = (ids >= i0) & (ids < i0 + shard.shape[0])
                bs[ids[own] - i0] = land[slots[own]]        # ... and their bias values
        loss_out.copy_(torch.tensor([loss / self.world]).reshape(loss_out.shape))   # the ranks' shares add up to the loss

    def apply_at(self, index, n_steps):
        if index + 1 < n_steps:
            self._gather(index + 1)


def simulate_global_steps(loss, world, skew=False, bias=None):
    """Single-process restatement: every rank's batch sees the pre-step tables; user rows are
    disjoint across ranks; item gradients are summed over ranks (rank order) and applied once.
    bias: the InteractionDot bias vectors go the same way (user values with their user, item values summed over ranks);
    returns (U, V, losses[, ub, cb])."""
    conf = _conf(loss, bias=bias)
    U, V = _tables()
    ub = np.zeros(NU, dtype=np.float32) if bias in ("user", "both") else None
    cb = np.zeros(NI, dtype=np.float32) if bias in ("item", "both") else None
    batches = _batches(world, skew=skew)
    lr = np.float32(conf.learn_rate)
    losses = []
    for t in range(STEPS):
        dV = np.zeros_like(V)
        dcb = np.zeros(NI, dtype=np.float32)
        newU = U.copy()
        new_ub = None if ub is None else ub.copy()
        step_losses = []
        for r in range(world):
            uid, cid = batches[r][t]
            o = so.StepOracle(U, V, conf, dtype=np.float32)
            if ub is not None:
                o.ub = ub.copy()
            if cb is not None:
                o.cb = cb.copy()
            step_losses.append(o.step_neg_shared(uid, cid))
            touched = np.unique(uid)
            newU[touched] = o.U[touched]
            if ub is not None:
                new_ub[touched] = o.ub[touched]
            if cb is not None:
                dcb += o.last_gb["cb"]
            dV += o.last["dV"]
        U = newU
        V = V - lr * dV
        if ub is not None:
            ub = new_ub
        if cb is not None:
            cb = cb - lr * dcb
        losses.append(step_losses)
    if bias:
        return U, V, losses, ub, cb
    return U, V, losses


def _worker(rank, world, port, loss, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conf = _conf(loss)
    U, V = _tables()
    us, isz = shard_size(NU, world), shard_size(NI, world)
    be = NumpyBackend(conf, U[rank * us:(rank + 1) * us].copy(), V[rank * isz:(rank + 1) * isz].copy(), world, isz,
                      rank * isz)
    tr = ShardedNegShared(be, NU, NI, B, D)
    losses = []
    for uid, cid in _batches(world)[rank]:
        l = tr.step(torch.from_numpy((uid - rank * us).astype(np.int32)), torch.from_numpy(cid.astype(np.int32)))
        losses.append(float(l[0]))
    ret[rank] = (be.U.copy(), be.V.copy(), losses)
    dist.barrier()
    dist.destroy_process_group()


def _worker_chunked(rank, world, port, loss, ret, skew=False, bias=None):
    """the same global steps through ShardedNegShared.run_epoch (plans + id all-to-all batched per chunk)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conf = _conf(loss, bias=bias)
    U, V = _tables()
    us, isz = shard_size(NU, world), shard_size(NI, world)
    be = ChunkedNumpyBackend(conf, U[rank * us:(rank + 1) * us].copy(), V[rank * isz:(rank + 1) * isz].copy(), world, isz,
                             rank * isz)
    if bias:
        be.set_bias(np.zeros(min(us, NU - rank * us), dtype=np.float32) if bias in ("user", "both") else None,
                    np.zeros(min(isz, NI - rank * isz), dtype=np.float32) if bias in ("item", "both") else None)
    tr = ShardedNegShared(be, NU, NI, B, D)
    bt = _batches(world, skew=skew)[rank]
    uid = torch.from_numpy(np.concatenate([(u - rank * us) for u, _ in bt]).astype(np.int32))
    cid = torch.from_numpy(np.concatenate([c for _, c in bt]).astype(np.int32))
    losses = torch.zeros(STEPS)
    tr.run_epoch(uid, cid, STEPS, losses)
    ret[rank] = (be.U.copy(), be.V.copy(), [float(x) for x in losses]) + ((getattr(be, "ub", None), getattr(be, "cb", None)) if bias else ())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_world2_matches_single_process_restatement(loss):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), loss, ret), nprocs=world, join=True)
    U, V, losses = simulate_global_steps(loss, world)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_ = ret[r]
        np.testing.assert_allclose(Ur, U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        for t in range(STEPS):
            assert abs(lr_[t] - losses[t][r]) < 1e-5 * max(1, abs(losses[t][r]))


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_world2_chunked_epoch_matches_single_process_restatement(loss):
    """run_epoch: one id all-to-all per chunk of steps, per step only rows / gradients -- same results"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_chunked, args=(world, _free_port(), loss, ret), nprocs=world, join=True)
    U, V, losses = simulate_global_steps(loss, world)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_ = ret[r]
        np.testing.assert_allclose(Ur, U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        for t in range(STEPS):
            assert abs(lr_[t] - losses[t][r]) < 1e-5 * max(1, abs(losses[t][r]))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_chunked_epoch_with_unequal_splits_and_an_empty_peer(loss, world):
    """the variable-split exchange with UNEQUAL per-peer counts and EMPTY peers (rank 0 asks nobody but
    itself; the last rank sends nine ids to shard 0 for each of its own, and one single row in step 1) -- the cases the
    one-rank RCCL run on the GPU box (counts = everything to oneself) cannot produce.  Same single-process restatement."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_chunked, args=(world, _free_port(), loss, ret, True), nprocs=world, join=True)
    U, V, losses = simulate_global_steps(loss, world, skew=True)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    # the premise: rank 0's requests never leave its shard, the last rank's splits are lopsided
    bt = _batches(world, skew=True)
    assert all((c < isz).all() for _, c in bt[0])
    last = np.concatenate([c for _, c in bt[world - 1]])
    assert (last < isz).mean() > 0.8 and len(np.unique(bt[world - 1][1][1])) == 1
    for r in range(world):
        Ur, Vr, lr_ = ret[r]
        np.testing.assert_allclose(Ur, U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        for t in range(STEPS):
            assert abs(lr_[t] - losses[t][r]) < 1e-5 * max(1, abs(losses[t][r]))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("loss,bias", [("skip-gram", "both"), ("mse", "item"), ("log-loss", "user")])
def test_weak_form_with_an_interaction_bias(loss, bias, world):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on the WEAK sharded form (round 6): the user bias
    is local like the user rows, the item bias values travel with the item rows and their gradients come back with the gradient
    rows; the owners add the ranks' values in rank order and take the SGD step.  Skewed batches (an empty peer, lopsided splits)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_chunked, args=(world, _free_port(), loss, ret, True, bias), nprocs=world, join=True)
    U, V, losses, ub, cb = simulate_global_steps(loss, world, skew=True, bias=bias)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_, ubr, cbr = ret[r]
        np.testing.assert_allclose(Ur, U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        for t in range(STEPS):
            assert abs(lr_[t] - losses[t][r]) < 1e-5 * max(1, abs(losses[t][r]))
        if ub is not None:
            np.testing.assert_allclose(ubr, ub[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
            assert np.abs(ub).max() > 1e-4
        else:
            assert ubr is None
        if cb is not None:
            np.testing.assert_allclose(cbr, cb[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
            assert np.abs(cb).max() > 1e-4
        else:
            assert cbr is None


def test_world1_equals_plain_step():
    """world_size 1 (no process group): the exchange degenerates to local copies."""
    conf = _conf("skip-gram")
    U, V = _tables()
    be = NumpyBackend(conf, U.copy(), V.copy(), 1, NI, 0)
    tr = ShardedNegShared(be, NU, NI, B, D)
    o = so.StepOracle(U, V, conf, dtype=np.float32)
    for uid, cid in _batches(1)[0]:
        l = tr.step(torch.from_numpy(uid.astype(np.int32)), torch.from_numpy(cid.astype(np.int32)))
        want = o.step_neg_shared(uid, cid)
        assert abs(float(l[0]) - want) < 1e-5 * max(1, abs(want))
    np.testing.assert_allclose(be.U, o.U, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(be.V, o.V, rtol=1e-5, atol=1e-6)


def _global_batches(seed=7):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(STEPS):
        uid = rs.randint(0, NU, B)
        uid[:5] = uid[0]                       # one user on several ranks' rows
        cid = rs.randint(0, 12 if _ == 1 else NI, B)   # step 1: heavy item duplication
        out.append((uid, cid))
    return out


def _worker_strong(rank, world, port, loss, ret, bias=None, optimizer="sgd"):
    from nncf_amd.sharded import ShardedGlobalBatch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conf = _conf(loss, bias, optimizer)
    U, V = _tables()
    us, isz = shard_size(NU, world), shard_size(NI, world)
    cls = ReplicatedNumpyBackend if (loss in ("log-loss", "max-margin") or bias) else StrongNumpyBackend
    be = cls(conf, U[rank * us:(rank + 1) * us].copy(), V[rank * isz:(rank + 1) * isz].copy(), world, rank, us, isz)
    tr = ShardedGlobalBatch(be, NU, NI, B, D)
    bt = _global_batches()
    uid = torch.from_numpy(np.concatenate([u for u, _ in bt]).astype(np.int32))
    cid = torch.from_numpy(np.concatenate([c for _, c in bt]).astype(np.int32))
    losses = torch.zeros(STEPS)
    tr.run_epoch(uid, cid, STEPS, losses)
    dist.all_reduce(losses)                    # a step's loss = the sum of the ranks' shares
    ret[rank] = (be.U.copy(), be.V.copy(), [float(x) for x in losses], tr.wire_bytes) + (
        (None if be.bu is None else be.bu.copy(), None if be.bi is None else be.bi.copy()) if bias else ())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("optimizer", ["adam", "sparse_adam"])
def test_replicated_global_batch_with_adam_over_two_ranks(optimizer, world):
    """pairwise loss + Adam on the strong form: the moment rows travel with the table rows of the replicated protocol (two more
    exchanges per table and step with the same counts), world 2 == the one-device step; Keras' dense Adam: the untouched rows of
    every shard take the zero-gradient step.  ShardedGlobalBatch / _Exchanger are the product's."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_strong, args=(world, _free_port(), "log-loss", ret, None, optimizer), nprocs=world, join=True)
    conf = _conf("log-loss", None, optimizer)
    U, V = _tables()
    orc = so.StepOracle(U, V, conf, dtype=np.float32)
    want = [orc.step_neg_shared(u, c) for u, c in _global_batches()]
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_, wire = ret[r]
        np.testing.assert_allclose(lr_, want, rtol=1e-5)
        np.testing.assert_allclose(Ur, orc.U[r * us:(r + 1) * us], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(Vr, orc.V[r * isz:(r + 1) * isz], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("loss,bias", [("skip-gram", "both"), ("log-loss", "user"), ("mse", "item")])
def test_global_batch_with_an_interaction_bias_over_two_ranks(loss, bias, world):
    """InteractionDot bias (interaction_dot.py:62-71,96-107) on the strong form: the replicated protocol with the bias values
    travelling next to the rows (a second exchange with the same counts), world 2 == the one-device step to 1e-5 -- rows, bias
    vectors and losses.  ShardedGlobalBatch / _Exchanger are the product's; the backend restates HipBackend on the oracle."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_strong, args=(world, _free_port(), loss, ret, bias), nprocs=world, join=True)
    conf = _conf(loss, bias)
    U, V = _tables()
    orc = so.StepOracle(U, V, conf, dtype=np.float32)
    want = [orc.step_neg_shared(u, c) for u, c in _global_batches()]
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_, wire, bu, bi = ret[r]
        np.testing.assert_allclose(lr_, want, rtol=1e-5)
        np.testing.assert_allclose(Ur, orc.U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, orc.V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        assert (bu is None) == (orc.ub is None) and (bi is None) == (orc.cb is None)
        if bu is not None:
            assert np.abs(orc.ub).max() > 1e-3
            np.testing.assert_allclose(bu, orc.ub[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        if bi is not None:
            assert np.abs(orc.cb).max() > 1e-3
            np.testing.assert_allclose(bi, orc.cb[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        assert wire > 0


@pytest.mark.parametrize("loss", ["skip-gram", "mse", "log-loss", "max-margin"])
@pytest.mark.parametrize("world", [2, 4])
def test_global_batch_over_ranks_equals_the_single_device_step(loss, world):
    """Strong form (SURVEY.md 8(e) as written): ONE batch of b links per step, rank g holds rows g, g + G, ... of the
    b x b matrix, users and items fetched from their owners, both gradients returned -- loss and both tables equal the
    one-device step (oracle, models/train_neg_shared.py:46-51 semantics) on the SAME batches to 1e-5.  Pairwise losses
    (log-loss, max-margin) run the REPLICATED form of HipBackend: all 2 b rows to every rank, the whole step everywhere, the
    owners write their rows back (ReplicatedNumpyBackend restates that protocol)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_strong, args=(world, _free_port(), loss, ret), nprocs=world, join=True)
    conf = _conf(loss)
    U, V = _tables()
    orc = so.StepOracle(U, V, conf, dtype=np.float32)
    want = [orc.step_neg_shared(u, c) for u, c in _global_batches()]
    us, isz = shard_size(NU, world), shard_size(NI, world)
    for r in range(world):
        Ur, Vr, lr_, wire = ret[r]
        np.testing.assert_allclose(lr_, want, rtol=1e-5)
        np.testing.assert_allclose(Ur, orc.U[r * us:(r + 1) * us], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vr, orc.V[r * isz:(r + 1) * isz], rtol=1e-5, atol=1e-6)
        assert wire > 0


def test_global_batch_world1_equals_plain_step():
    from nncf_amd.sharded import ShardedGlobalBatch
    conf = _conf("skip-gram")
    U, V = _tables()
    be = StrongNumpyBackend(conf, U.copy(), V.copy(), 1, 0, NU, NI)
    tr = ShardedGlobalBatch(be, NU, NI, B, D)
    bt = _global_batches()
    uid = torch.from_numpy(np.concatenate([u for u, _ in bt]).astype(np.int32))
    cid = torch.from_numpy(np.concatenate([c for _, c in bt]).astype(np.int32))
    losses = torch.zeros(STEPS)
    tr.run_epoch(uid, cid, STEPS, losses)
    orc = so.StepOracle(U, V, conf, dtype=np.float32)
    want = [orc.step_neg_shared(u, c) for u, c in bt]
    np.testing.assert_allclose(losses.numpy(), want, rtol=1e-5)
    np.testing.assert_allclose(be.U, orc.U, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(be.V, orc.V, rtol=1e-5, atol=1e-6)


def test_chunked_world1_numpy_equals_plain_step():
    conf = _conf("skip-gram")
    U, V = _tables()
    be = ChunkedNumpyBackend(conf, U.copy(), V.copy(), 1, NI, 0)
    tr = ShardedNegShared(be, NU, NI, B, D)
    bt = _batches(1)[0]
    uid = torch.from_numpy(np.concatenate([u for u, _ in bt]).astype(np.int32))
    cid = torch.from_numpy(np.concatenate([c for _, c in bt]).astype(np.int32))
    losses = torch.zeros(STEPS)
    tr.run_epoch(uid, cid, STEPS, losses)
    o = so.StepOracle(U, V, conf, dtype=np.float32)
    want = [o.step_neg_shared(u, c) for u, c in bt]
    np.testing.assert_allclose(losses.numpy(), want, rtol=1e-5)
    np.testing.assert_allclose(be.U, o.U, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(be.V, o.V, rtol=1e-5, atol=1e-6)


def test_partition_links_by_user():
    rs = np.random.RandomState(0)
    train = np.stack([rs.randint(0, 37, 300), rs.randint(0, 50, 300), np.ones(300, dtype=np.int64)], 1)
    parts = partition_links_by_user(train, 37, 4)
    assert sum(len(p) for p in parts) == 300
    us = shard_size(37, 4)
    for r, p in enumerate(parts):
        assert np.all(p[:, 0] // us == r)


# ------------------------------------------------------------------------------------------
# basic_embedding, data parallel: DataParallelTower over gloo with an oracle-backed replica
# ------------------------------------------------------------------------------------------
from nncf_amd.sharded import DataParallelTower  # noqa: E402

TW_ITEMS, TW_L, TW_VOCAB = 50, 6, 30


def _tower_setup(seed=3):
    rs = np.random.RandomState(seed)
    C = rs.randint(0, TW_VOCAB, (TW_ITEMS, TW_L))
    W = rs.uniform(-.5, .5, (TW_VOCAB, D)).astype(np.float32)
    Wd = rs.uniform(-.5, .5, (D, D)).astype(np.float32)
    U = rs.uniform(-.5, .5, (NU, D)).astype(np.float32)
    return C, W, Wd, U


TW_F, TW_FL = 5, 3   # the CNN variant: Convolution1D(5 filters, length 3) in front of the same tail


def _tower_oracle(conf, cnn=False):
    C, W, Wd, U = _tower_setup()
    z, o = (lambda n: np.zeros(n, np.float32)), (lambda n: np.ones(n, np.float32))
    if cnn:
        rs = np.random.RandomState(9)
        Wc = rs.uniform(-.5, .5, (TW_FL, D, TW_F)).astype(np.float32)
        Wd = rs.uniform(-.5, .5, (TW_F, D)).astype(np.float32)
        to = so.CnnTowerOracle(C, W, Wc, z(TW_F), o(TW_F), z(TW_F), Wd, z(D), o(D), z(D), conf.learn_rate, dtype=np.float32)
    else:
        to = so.TowerOracle(C, W, Wd, z(D), o(D), z(D), conf.learn_rate, dtype=np.float32)
    to.export_only = True
    return so.StepOracle(U, None, conf, dtype=np.float32, tower=to), to


class NumpyTowerReplica(object):
    """Test double of HipTowerReplica: oracle arithmetic, gradients exported to a flat CPU tensor."""

    def __init__(self, conf, cnn=False):
        self.orc, self.to = _tower_oracle(conf, cnn)
        self.grad = torch.zeros(self.to.W.size + self.to.Wd.size + 384 + ((self.to.Wc.size + 384) if cnn else 0))

    def step(self, uid, cid, loss_out=None):
        loss = self.orc.step_neg_shared(uid.numpy(), cid.numpy())
        self.grad.copy_(torch.from_numpy(self.to.flat_grads()))
        return torch.tensor([loss])

    def apply(self, scale=1.0):
        self.to.apply_flat(self.grad.numpy(), scale)


def _tower_batches(world, seed=4):
    rs = np.random.RandomState(seed)
    us = shard_size(NU, world)
    return [[(rs.randint(r * us, min(NU, (r + 1) * us), B), rs.randint(0, TW_ITEMS, B)) for _ in range(STEPS)]
            for r in range(world)]


def _dp_worker(rank, world, port, ret, cnn):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conf = so.OracleConf(batch_size_p=B, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.1,
                         model_choice="basic_embedding")
    rep = NumpyTowerReplica(conf, cnn)
    tr = DataParallelTower(rep)
    losses = []
    for uid, cid in _tower_batches(world)[rank]:
        losses.append(float(tr.step(torch.from_numpy(uid), torch.from_numpy(cid))[0]))
    ret[rank] = (rep.orc.U.copy(), {k: getattr(rep.to, k).copy() for k in rep.to.PARAMS}, losses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cnn", [False, True], ids=["mean_pool", "cnn"])
def test_data_parallel_tower_world2(cnn):
    """Two ranks, one all-reduce per step: replicas stay identical and equal a single-process
    restatement (every rank's gradient from the same pre-step parameters, summed, applied once);
    user rows are touched only by their owner.  Both text towers (the CNN tower's flat buffer also
    carries the convolution and its BatchNorm)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), ret, cnn), nprocs=world, join=True)
    conf = so.OracleConf(batch_size_p=B, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.1,
                         model_choice="basic_embedding")
    reps = [NumpyTowerReplica(conf, cnn) for _ in range(world)]
    batches = _tower_batches(world)
    losses = [[] for _ in range(world)]
    for t in range(STEPS):
        flats = []
        for r in range(world):
            losses[r].append(float(reps[r].step(torch.from_numpy(batches[r][t][0]), torch.from_numpy(batches[r][t][1]))[0]))
            flats.append(reps[r].grad.numpy().copy())
        total = np.sum(flats, axis=0)
        for r in range(world):
            reps[r].to.apply_flat(total)
    us = shard_size(NU, world)
    for r in range(world):
        Ur, params, lr_ = ret[r]
        np.testing.assert_allclose(lr_, losses[r], rtol=1e-5)
        for k, v in params.items():
            np.testing.assert_allclose(v, getattr(reps[r].to, k), rtol=1e-5, atol=1e-6, err_msg=k)
            np.testing.assert_array_equal(v, ret[0][1][k])                  # replicas bit-identical
        own = slice(r * us, (r + 1) * us)
        np.testing.assert_allclose(Ur[own], reps[r].orc.U[own], rtol=1e-5, atol=1e-6)
        other = np.setdiff1d(np.arange(NU), np.arange(NU)[own])
        np.testing.assert_array_equal(Ur[other], _tower_setup()[3][other])   # never touched here
