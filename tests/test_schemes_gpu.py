"""GPU: the remaining train schemes (presample / group_sample / reverse through the row-wise
kernels, sampled_neg_shared through the per-positive kernels) against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from oracle.sampler_oracle import OracleSampler  # noqa: E402
from nncf_amd import synthetic  # noqa: E402
from nncf_amd.conf import get_conf  # noqa: E402
from nncf_amd.data_utils import degree_dist, get_data, group_shuffle_train  # noqa: E402
from nncf_amd.model_framework import get_model  # noqa: E402


@pytest.fixture(scope="module")
def small_data():
    return synthetic.make_citeulike_like(users=400, items=300, train_links=6000, seed=21)


def _conf(loss, b=32, k=3, d=16, lr=0.2, shuffle_st="by_item_chop", epochs=2):
    pd = {"reset_after_getconf": True, "max_epoch": epochs, "loss": loss, "num_negatives": k,
          "neg_loss_weight": 8 if loss == "mse" else 16, "loss_gamma": 0.1 if loss == "max-margin" else 10,
          "learn_rate": lr, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": b, "user_dim": d,
          "item_dim": d, "chop_size": 4, "group_shuffling_trick": True, "interaction_bias": None, "u_reg": 1e-4,
          "shuffle_st": shuffle_st}
    return get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="mf")


def _oconf(conf):
    return so.OracleConf(batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
                         loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
                         learn_rate=conf.learn_rate, emb_normalization=conf.emb_normalization, model_choice="mf")


@pytest.mark.parametrize("loss", ["skip-gram", "mse"])
def test_rowwise_train_on_batch_matches_oracle(small_data, loss):
    conf = _conf(loss)
    helper = get_data(dict(small_data), conf, reverse_samping=True, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=4)
    eng = md["model"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oconf(conf), dtype=np.float32)
    n = conf.batch_size_p * (1 + conf.num_negatives)
    rs = np.random.RandomState(0)
    for step in range(3):
        uid, cid = rs.randint(0, 400, n), rs.randint(0, 300, n)
        uid[5] = uid[60]
        cid[7] = cid[90]
        y = np.where(rs.rand(n) < 0.3, 1, -1 if loss == "skip-gram" else 0)
        want = orc.step_original(uid, cid, y)
        got = md["model"].train_on_batch([uid, cid], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want))
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)
    conf2 = _conf("log-loss")
    md2 = get_model(conf2, helper, "mf", seed=4)
    with pytest.raises(ValueError):  # pairwise losses need the positives-first layout
        md2["model"].train_on_batch([uid, cid], [np.where(y == 1, 1, 0)])


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss", "max-margin", "mse"])
def test_sampled_neg_shared_matches_oracle(small_data, loss):
    conf = _conf(loss, k=5)
    helper = get_data(dict(small_data), conf, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=5)
    eng = md["model"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oconf(conf), dtype=np.float32)
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(1)
    for step in range(3):
        uid = np.concatenate([rs.randint(0, 400, b), np.zeros(k, dtype=int)])
        cid = np.concatenate([rs.randint(0, 300, b), rs.randint(0, 300, k)])
        want = orc.step_sampled_neg_shared(uid, cid)
        got = md["model_sampled_neg_shared"].train_on_batch([uid, cid], None)
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=1e-5, atol=2e-6)


def _ref_epoch_rows(scheme, conf, train, epoch, iidx, smp_i, smp_u):
    """The reference trainers' batch assembly, written out (train_presample.py:44-85,
    train_group_sample.py:64-83 with the by-user fix, train_reverse.py:36,57-63)."""
    b, k = conf.batch_size_p, conf.num_negatives
    neg_sign = -1 if conf.loss == "skip-gram" else 0
    if scheme == "presample":
        train_n = train.repeat(k, axis=0)
        train_n[:, 1] = smp_i.sample_batch(train_n.shape[0])
        train_n[:, 2] = neg_sign
        rows = np.vstack((train, train_n))
        rows = group_shuffle_train(rows, by="item", chop=conf.chop_size, iidx=iidx["item"])
        return train, rows
    if scheme == "group_sample":
        train = group_shuffle_train(train, by="item", chop=conf.chop_size, iidx=iidx["item"])
    else:
        np.random.shuffle(train)
    out = []
    bb = 0
    while epoch > 0 and bb + b <= len(train):
        bp = train[bb:bb + b]
        bn = bp.repeat(k, axis=0)
        bn[:, 0] = smp_u.sample_batch(k * b)
        bn[:, 2] = neg_sign
        out.append(np.vstack((bp, bn)))
        bb += b
    return train, (np.vstack(out) if out else np.zeros((0, 3), dtype=np.int64))


@pytest.mark.parametrize("scheme", ["presample", "group_sample", "reverse", "sampled_neg_shared"])
def test_scheme_trainers_match_oracle_loops(small_data, scheme):
    conf = _conf("skip-gram", shuffle_st="by_item_chop")
    data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in small_data.items()}
    helper = get_data(data, conf, reverse_samping=True, rand_seed=99)
    md = get_model(conf, helper, "mf", seed=6)
    eng = md["model"].engine
    U0, V0 = eng.U.cpu().numpy().copy(), eng.V.cpu().numpy().copy()
    mod = __import__("nncf_amd.train_%s" % scheme, fromlist=["Trainer"])
    np.random.seed(7)
    tr = mod.Trainer(md, conf, helper)
    tr.train(None)

    orc = so.StepOracle(U0, V0, _oconf(conf), dtype=np.float32)
    train = small_data["train"].copy()
    smp_i = OracleSampler(degree_dist(train, 1, "unigram"), 1.0, 99)
    smp_u = OracleSampler(degree_dist(train, 0, "unigram"), 1.0, 99)
    iidx = {"item": np.arange(train[:, 1].max() + 1), "user": np.arange(train[:, 0].max() + 1)}
    b, k = conf.batch_size_p, conf.num_negatives
    n = b * (1 + k)
    np.random.seed(7)
    costs = []
    for epoch in range(conf.max_epoch + 1):
        cost, it = 0., 0
        if scheme == "sampled_neg_shared":
            train = group_shuffle_train(train, by="item", chop=conf.chop_size, iidx=iidx["item"])
            bb = 0
            while epoch > 0 and bb < len(train):
                it += 1                           # train_sampled_neg_shared.py:41-45
                if bb + b > len(train):
                    break
                bt = train[bb:bb + b]
                negs = smp_i.sample_batch(k)
                cost += orc.step_sampled_neg_shared(np.concatenate([bt[:, 0], np.zeros(k, dtype=int)]),
                                                    np.concatenate([bt[:, 1], negs]))
                bb += b
        else:
            train, rows = _ref_epoch_rows(scheme, conf, train, epoch, iidx, smp_i, smp_u)
            bb = 0
            while epoch > 0 and bb < len(rows):
                it += 1                           # train_presample.py:88-93 and siblings
                if bb + n > len(rows):
                    break
                r = rows[bb:bb + n]
                cost += orc.step_original(r[:, 0], r[:, 1], r[:, 2])
                bb += n
            if scheme in ("group_sample", "reverse") and it and len(train) % b and len(rows) % n == 0:
                it += 1                           # those loops walk the link list (train_group_sample.py:69-74)
        costs.append(cost / it if it else -1)
    got = [h[2] for h in tr.history]
    for g, w in zip(got, costs):
        assert abs(g - w) <= 1e-5 * max(1, abs(w)), (scheme, got, costs)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("scheme", ["group_neg_shared", "group_sample"])
def test_group_sampler_trainers_match_oracle_loops(small_data, scheme):
    """group_shuffling_trick = False: every batch comes from GroupSampler (HIP sampler streams +
    NumPy member draws).  The trainer against the reference's loop written out on the oracle with
    a CPU GroupSampler fed by the checker sampler: same seeds -> same batches -> same cost."""
    import random
    import importlib
    from nncf_amd.data_utils import GroupSampler
    loss = "log-loss" if scheme == "group_neg_shared" else "skip-gram"
    conf = _conf(loss, epochs=2)
    conf.group_shuffling_trick = False
    if scheme == "group_neg_shared":
        conf.num_negatives = 0
    data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in small_data.items()}
    helper = get_data(data, conf, reverse_samping=True, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=4)
    eng = md["model"].engine
    oc = _oconf(conf)
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), oc, dtype=np.float32)
    Trainer = importlib.import_module("nncf_amd.train_" + scheme).Trainer
    np.random.seed(11)
    random.seed(12)
    tr = Trainer(md, conf, helper)
    # the reference seeds these samplers from time(NULL) (rand_seed = 0, nodesampler.cpp:59-60);
    # pin their LCG states for the comparison
    tr.group_sampler.group_sampler.__self__.state = 0
    tr.group_sampler.member_neg_sampler.__self__.state = 0
    tr.train(None)
    # the same loop on the oracle
    def factory(ratings, neg_dist, power, column, batch_mode):
        return OracleSampler(degree_dist(ratings, column, neg_dist), power, 0).sample_batch
    neg_sign = -1 if loss == "skip-gram" else 0
    gs = GroupSampler(small_data["train"], group_by="item", chop=conf.chop_size, neg_dist=conf.neg_dist, neg_sign=neg_sign,
                      sampler_factory=factory)
    np.random.seed(11)
    random.seed(12)
    b, k = conf.batch_size_p, conf.num_negatives
    n_steps = len(small_data["train"]) // b
    for epoch in (1, 2):
        cost = 0.0
        for _ in range(n_steps):
            if scheme == "group_neg_shared":
                rows = gs.sample(b)
                cost += orc.step_group_neg_shared(rows[:, 0], rows[:, 1])
            else:
                rows = gs.sample_with_negs(b, k)
                cost += orc.step_original(rows[:, 0], rows[:, 1], rows[:, 2])
        got = tr.history[epoch][2]
        it = n_steps + (1 if len(small_data["train"]) % b else 0)   # the reference counts the broken-off tail iteration
        assert abs(got - cost / it) <= 1e-5 * max(1, abs(cost / it)), (epoch, got, cost / it)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)


def test_group_sample_trainer_with_device_sample_with_negs(small_data):
    """train_scheme = group_sample, group_shuffling_trick = False, conf.device_shuffle = True: every epoch's batches come
    from GroupSampler.sample_with_negs_device (no host loop, no upload) -- the trainer against the oracle loop fed with the
    device sampler's restatement (oracle/shuffle_oracle.py) on the checker samplers' streams."""
    import importlib
    from oracle.shuffle_oracle import group_sample_with_negs_rows
    conf = _conf("skip-gram", epochs=2)
    conf.group_shuffling_trick = False
    conf.device_shuffle = True
    data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in small_data.items()}
    helper = get_data(data, conf, reverse_samping=True, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=4)
    eng = md["model"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oconf(conf), dtype=np.float32)
    tr = importlib.import_module("nncf_amd.train_group_sample").Trainer(md, conf, helper)
    gs = tr.group_sampler
    gs.group_sampler.__self__.state = 0
    gs.member_neg_sampler.__self__.state = 0
    tr.train(None)
    b, k, chop = conf.batch_size_p, conf.num_negatives, conf.chop_size
    train = small_data["train"]
    n_steps = len(train) // b
    gcap, ncap = gs.negs_caps(b, k)
    g_or = OracleSampler(degree_dist(train, 1, "unigram"), 1.0, 0)
    m_or = OracleSampler(degree_dist(train, 0, conf.neg_dist), 0.75, 0)
    ptr, mem = [t.cpu().numpy() for t in gs._dev_csr]
    n = b * (1 + k)
    for epoch in (1, 2):
        u, c, y, used = group_sample_with_negs_rows(g_or.sample_batch(n_steps * gcap), m_or.sample_batch(n_steps * ncap), gcap, ncap,
                                                    b, k, chop, ptr, mem, None, epoch + 1, neg_sign=-1)
        assert used.min() > 0
        cost = sum(orc.step_original(u[t * n:(t + 1) * n], c[t * n:(t + 1) * n], y[t * n:(t + 1) * n]) for t in range(n_steps))
        it = n_steps + (1 if len(train) % b else 0)
        got = tr.history[epoch][2]
        assert abs(got - cost / it) <= 1e-5 * max(1, abs(cost / it)), (epoch, got, cost / it)
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-4, atol=2e-5)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "12"))))
def test_rowwise_random_batches_match_oracle(small_data, seed):
    """The row-wise kernels (presample / group_sample / reverse feed them arbitrary (user, item, label) rows): random b, k, d, both
    pointwise losses, label mixes from all-negative to all-positive, ids drawn from 1 ... 400 users and 1 ... 300 items (every
    row a duplicate at one end) -- three steps against the float64 oracle."""
    rs = np.random.RandomState(19000 + seed)
    loss = str(rs.choice(["skip-gram", "mse"]))
    b, k, d = int(rs.choice([1, 8, 32, 100, 512])), int(rs.choice([1, 3, 10])), int(rs.choice([4, 16, 50, 100, 128]))
    conf = _conf(loss, b=b, k=k, d=d, lr=0.05)
    helper = get_data(dict(small_data), conf, reverse_samping=True, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=4)
    eng = md["model"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oconf(conf), dtype=np.float64)
    n = b * (1 + k)
    nu, ni = int(rs.choice([1, 2, 30, 400])), int(rs.choice([1, 2, 30, 300]))
    ppos = float(rs.choice([0.0, 0.1, 0.5, 1.0]))
    for step in range(3):
        uid, cid = rs.randint(0, nu, n), rs.randint(0, ni, n)
        y = np.where(rs.rand(n) < ppos, 1, -1 if loss == "skip-gram" else 0)
        want = orc.step_original(uid, cid, y)
        got = md["model"].train_on_batch([uid, cid], [y])
        info = (loss, b, k, d, nu, ni, ppos, step)
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (info, got, want)
        amp = max(1.0, n / (512.0 * min(nu, ni)))   # the terms that meet in one row
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=4e-6 * amp, err_msg=str(info))
        np.testing.assert_allclose(eng.V.cpu().numpy(), orc.V, rtol=2e-5, atol=4e-6 * amp, err_msg=str(info))
