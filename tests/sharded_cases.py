"""Shared by tests/test_sharded_two_ranks_gpu.py and its rank processes (tests/sharded_rank_worker.py): the seeded inputs
of the multi-rank cases and their single-process restatements on the oracle.  Test infrastructure only."""
import numpy as np

from oracle import step_oracle as so

B, D, NU, NI, STEPS = 64, 20, 300, 500, 70        # 70 steps = one full chunk of 64 and a partial one
LR, UREG, LAMBDA = 0.1, 1e-3, 16.0


def shard_size(count, world):
    return (int(count) + world - 1) // world


LR_ADAM = 0.01


def conf(loss="skip-gram", model="mf", optimizer="sgd", bias=None):
    return so.OracleConf(batch_size_p=B, num_negatives=0, loss=loss, loss_gamma=10., neg_loss_weight=LAMBDA, u_reg=UREG,
                         learn_rate=LR if optimizer == "sgd" else LR_ADAM, model_choice=model, optimizer=optimizer,
                         interaction_bias=bias)


def engine_kwargs(loss="skip-gram", optimizer="sgd", bias=None):
    # the pairwise losses normalise the user rows by default, as the reference's conf does (basic_embedding_conf.py:52-56;
    # OracleConf follows it), id-table items stay raw (model_framework.py:85-88)
    kw = dict(user_count=NU, item_count=NI, dim=D, batch_size_p=B, num_negatives=0, loss=loss, loss_gamma=10.,
              neg_loss_weight=LAMBDA, u_reg=UREG, learn_rate=LR if optimizer == "sgd" else LR_ADAM,
              normalize_user=conf(loss).normalize_user, normalize_item=False)
    if optimizer != "sgd":
        kw["optimizer"] = optimizer      # HipBackend: the engine's and the shards' Adam state
    if bias:
        kw["interaction_bias"] = bias    # InteractionDot bias (interaction_dot.py:62-71): the strong form runs replicated
    return kw


def tables(seed=0):
    rs = np.random.RandomState(seed)
    return (rs.uniform(-.5, .5, (NU, D)).astype(np.float32), rs.uniform(-.5, .5, (NI, D)).astype(np.float32))


def weak_batches(world, seed=1, skew=False):
    """per rank: STEPS batches of B links whose users lie in the rank's range; odd steps repeat few items.
    skew: the variable-split cases a one-rank run cannot produce -- rank 0 asks ONLY its own item shard
    (an EMPTY request to every other peer, every step of every chunk), the last rank asks ~nine ids of shard 0 for each of its
    own (lopsided splits), and in steps 5 and 66 (one in each chunk) one single item B times (a one-row exchange)."""
    rs = np.random.RandomState(seed)
    us, isz = shard_size(NU, world), shard_size(NI, world)
    out = []
    for r in range(world):
        lo, hi = r * us, min(NU, (r + 1) * us)
        steps = []
        for t in range(STEPS):
            cid = rs.randint(0, 40 if t % 2 else NI, B)
            if skew and r == 0:
                cid = rs.randint(0, 40 if t % 2 else isz, B)
            elif skew and r == world - 1:
                cid = np.where(rs.rand(B) < 0.9, rs.randint(0, isz, B), rs.randint(r * isz, NI, B))
                if t in (5, 66):
                    cid = np.full(B, 7)
            steps.append((rs.randint(lo, hi, B), cid))
        out.append(steps)
    return out


def global_batches(seed=7):
    rs = np.random.RandomState(seed)
    out = []
    for t in range(STEPS):
        uid = rs.randint(0, NU, B)
        uid[:5] = uid[0]                       # one user on several ranks' rows
        out.append((uid, rs.randint(0, 12 if t % 3 == 1 else NI, B)))
    return out


def simulate_weak(loss, world, skew=False, bias=None):
    """every rank's batch sees the pre-step tables; user rows are disjoint across ranks; item gradients are summed over
    ranks (rank order) and applied once.  bias: the InteractionDot bias vectors likewise (user values with their user, the item
    values' gradients summed over ranks); returns (U, V, losses[, ub, cb])."""
    c = conf(loss, bias=bias)
    U, V = tables()
    ub = np.zeros(NU, dtype=np.float32) if bias in ("user", "both") else None
    cb = np.zeros(NI, dtype=np.float32) if bias in ("item", "both") else None
    bt = weak_batches(world, skew=skew)
    lr = np.float32(c.learn_rate)
    losses = []
    for t in range(STEPS):
        dV = np.zeros_like(V)
        dcb = np.zeros(NI, dtype=np.float32)
        newU = U.copy()
        new_ub = None if ub is None else ub.copy()
        row = []
        for r in range(world):
            uid, cid = bt[r][t]
            o = so.StepOracle(U, V, c, dtype=np.float32)
            if ub is not None:
                o.ub = ub.copy()
            if cb is not None:
                o.cb = cb.copy()
            row.append(o.step_neg_shared(uid, cid))
            touched = np.unique(uid)
            newU[touched] = o.U[touched]
            if ub is not None:
                new_ub[touched] = o.ub[touched]
            if cb is not None:
                dcb += o.last_gb["cb"]
            dV += o.last["dV"]
        U, V = newU, V - lr * dV
        if ub is not None:
            ub = new_ub
        if cb is not None:
            cb = cb - lr * dcb
        losses.append(row)
    if bias:
        return U, V, np.asarray(losses), ub, cb
    return U, V, np.asarray(losses)


def simulate_strong(loss, optimizer="sgd", bias=None):
    """the one-device step (models/train_neg_shared.py:46-51 semantics) on the global batches, float64; with an interaction
    bias: (U, V, losses, user bias or None, item bias or None)"""
    c = conf(loss, optimizer=optimizer, bias=bias)
    U, V = tables()
    o = so.StepOracle(U, V, c, dtype=np.float64)
    losses = [o.step_neg_shared(u, cc) for u, cc in global_batches()]
    if bias:
        return o.U, o.V, np.asarray(losses), o.ub, o.cb
    return o.U, o.V, np.asarray(losses)


# ---- data-parallel text tower --------------------------------------------------------------------------
TW = dict(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)
TW_PD = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
         "loss_gamma": 10, "learn_rate": 0.05, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
         "user_dim": 24, "item_dim": 24, "word_dim": 24, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.,
         "optimizer": "sgd"}


def tower_streams(train, users, world, seed=2):
    """per rank: STEPS * 64 links of its user range (with replacement)"""
    rs = np.random.RandomState(seed)
    us = shard_size(users, world)
    owner = train[:, 0] // us
    out = []
    for r in range(world):
        mine = train[owner == r]
        out.append(mine[rs.choice(len(mine), STEPS * 64, replace=True)])
    return out
