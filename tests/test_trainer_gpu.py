"""GPU: the host mirror (get_model / Trainer / Evaluator) end to end against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd import synthetic  # noqa: E402
from nncf_amd.conf import get_conf  # noqa: E402
from nncf_amd.data_utils import get_data, group_shuffle_train  # noqa: E402
from nncf_amd.model_framework import get_model  # noqa: E402


@pytest.fixture(scope="module")
def small_data():
    return synthetic.make_citeulike_like(users=600, items=500, train_links=12000, seed=11)


def _conf(loss, scheme, b=64, k=4, d=24, lr=0.3, epochs=2, chop=4):
    pd = {"reset_after_getconf": True, "max_epoch": epochs, "loss": loss, "num_negatives": k if scheme == "original" else 0,
          "neg_loss_weight": 8 if loss == "mse" else 16, "loss_gamma": 0.1 if loss == "max-margin" else 10,
          "learn_rate": lr, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": b, "user_dim": d,
          "item_dim": d, "chop_size": chop, "group_shuffling_trick": True, "interaction_bias": None, "u_reg": 1e-4}
    return get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="mf")


def _oracle_conf(conf):
    return so.OracleConf(batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
                         loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
                         learn_rate=conf.learn_rate, emb_normalization=conf.emb_normalization, model_choice="mf")


def test_model_dict_surface_and_train_on_batch(small_data):
    conf = _conf("skip-gram", "original")
    helper = get_data(dict(small_data), conf, rand_seed=5)
    md = get_model(conf, helper, "mf")
    assert sorted(md.keys()) == sorted(["model", "model_neg_shared", "model_group_neg_shared",
                                        "model_sampled_neg_shared", "model_user_emb", "model_item_emb",
                                        "model_pred_pairs", "model_all_loss", "model_neg_shared_all_loss"])
    eng = md["model"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oracle_conf(conf), dtype=np.float32)
    b, k = conf.batch_size_p, conf.num_negatives
    train = helper.data["train"]
    batch_p = train[:b]
    negs = helper.sampler_dict["sample_batch"](b * k)
    uid, cid, y = so.assemble_original(batch_p, negs, conf.loss)
    # loss-monitor models (SGD(0.)): evaluate every loss on the same batch, parameters untouched
    U0 = eng.U.clone()
    for lname, m in md["model_all_loss"].items():
        oc = _oracle_conf(conf)
        oc.loss = lname
        want = so.StepOracle(orc.U, orc.V, oc, dtype=np.float32).step_original(
            uid, cid, np.where(np.arange(len(uid)) < b, 1, -1 if lname == "skip-gram" else 0), update=False)
        got = m.train_on_batch([uid, cid], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), lname
    assert torch.equal(U0, eng.U)
    got = md["model"].train_on_batch([uid, cid], [y])  # int64 host arrays, like the reference trainer
    want = orc.step_original(uid, cid, y)
    assert abs(got - want) <= 1e-5 * max(1, abs(want))
    np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
    with pytest.raises(ValueError):
        md["model"].train_on_batch([uid[:-1], cid[:-1]], [y[:-1]])  # not batch_size_p * (1 + k) rows
    with pytest.raises(ValueError):
        md["model_group_neg_shared"].train_on_batch([uid[:b - 1], cid[:b - 1]], [y[:b - 1]])
    # predict_on_batch of the neg_shared head: (p, q) scores
    pm = md["model_neg_shared"].predict_on_batch([np.arange(7), np.arange(9)])
    np.testing.assert_allclose(pm, orc.predict_neg_shared(np.arange(7), np.arange(9)), rtol=1e-5, atol=1e-6)


def test_evaluator_whole_matches_oracle(small_data):
    conf = _conf("log-loss", "neg_shared")
    helper = get_data(dict(small_data), conf, rand_seed=5)
    md = get_model(conf, helper, "mf", seed=3)
    from nncf_amd.objectives import Evaluator
    ev = Evaluator(helper, helper.data_spec, conf)
    eng = md["model_neg_shared"].engine
    orc = so.StepOracle(eng.U.cpu().numpy(), eng.V.cpu().numpy(), _oracle_conf(conf), dtype=np.float64)
    _, uh, _, _ = orc._user_side(np.arange(helper.data_spec.user_count))
    _, vh, _ = orc._item_side(np.arange(helper.data_spec.item_count))
    for which, links in (("train", helper.data["train"]), ("test", helper.data["test"])):
        got = ev._whole(md["model_neg_shared"], which, 50)
        want = so.evaluate_whole(uh, vh, links[:, :2], links[:, 1], 50)
        assert got["users"] == want["users"]
        for name in ("recall", "map", "precision"):
            assert abs(got["%s@50" % name] - want[name]) < 1e-6, (which, name, got, want)


@pytest.mark.parametrize("scheme,loss", [("original", "skip-gram"), ("neg_shared", "skip-gram"),
                                         ("neg_shared", "log-loss"), ("group_neg_shared", "log-loss"),
                                         ("group_neg_shared", "max-margin"), ("original", "mse")])
def test_trainer_epochs_match_oracle_loop(small_data, scheme, loss):
    """Trainer.train() (device epoch loop) vs the reference's loop written out on the oracle,
    same NumPy seed => same shuffles, same sampler state => same negatives."""
    conf = _conf(loss, scheme, epochs=2)
    data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in small_data.items()}
    helper = get_data(data, conf, rand_seed=777)
    md = get_model(conf, helper, "mf", seed=1)
    eng = md["model"].engine
    U0, V0 = eng.U.cpu().numpy().copy(), eng.V.cpu().numpy().copy()
    mod = __import__("nncf_amd.train_%s" % scheme, fromlist=["Trainer"])
    np.random.seed(99)
    trainer = mod.Trainer(md, conf, helper)
    trainer.train("whole")

    # reference loop on the oracle
    from oracle.sampler_oracle import OracleSampler
    from nncf_amd.data_utils import degree_dist
    orc = so.StepOracle(U0, V0, _oracle_conf(conf), dtype=np.float32)
    train = small_data["train"].copy()
    b, k = conf.batch_size_p, conf.num_negatives
    smp = OracleSampler(degree_dist(train, 1, "unigram"), 1.0, 777) if scheme == "original" else None
    iidx = np.arange(train[:, 1].max() + 1)
    np.random.seed(99)
    costs = []
    for epoch in range(conf.max_epoch + 1):
        if scheme == "group_neg_shared":
            train = group_shuffle_train(train, by="item", chop=conf.chop_size, iidx=iidx)
        else:
            np.random.shuffle(train)
        cost, it = 0., 0
        bb = 0
        while epoch > 0 and bb < len(train):
            it += 1                               # train_original.py:44-48: counted before the ragged-tail check
            if bb + b > len(train):
                break
            bt = train[bb:bb + b]
            if scheme == "original":
                u, c, y = so.assemble_original(bt, smp.sample_batch(k * b), conf.loss)
                cost += orc.step_original(u, c, y)
            elif scheme == "neg_shared":
                cost += orc.step_neg_shared(bt[:, 0], bt[:, 1])
            else:
                cost += orc.step_group_neg_shared(bt[:, 0], bt[:, 1])
            bb += b
        costs.append(cost / it if it else -1)
    got = [h[2] for h in trainer.history]
    assert got[0] == -1 and costs[0] == -1          # epoch 0 trains nothing (train_original.py:34,43)
    for g, w in zip(got[1:], costs[1:]):
        assert abs(g - w) <= 1e-5 * max(1, abs(w)), (got, costs)
    # trained tables: bounds from the number of summed terms (tests/tolerances.py), not flat constants.  Both sides are fp32 here
    # (the oracle loop runs in float32 like the reference), each with its own summation order: twice the fp32-vs-float64 bound.
    from tests.tolerances import sum_bound
    steps_run = conf.max_epoch * (len(small_data["train"]) // b)
    for name, got_t, want_t, init_t in (("U", eng.U.cpu().numpy(), orc.U, U0), ("V", eng.V.cpu().numpy(), orc.V, V0)):
        bound = 2.0 * sum_bound(want_t, init_t, terms=b * (1 + k), updates=steps_run)
        err = np.abs(got_t.astype(np.float64) - want_t.astype(np.float64))
        assert (err <= bound).all(), "%s: worst |err| / bound = %.3f" % (name, float((err / bound).max()))
    # Recall@50 of the trained model: GPU evaluator on GPU tables vs oracle evaluator on oracle tables.
    # Rank flips from 1e-6-level table differences move recall by O(1/users); tolerance says so.
    ocf = _oracle_conf(conf)
    o64 = so.StepOracle(orc.U, orc.V, ocf, dtype=np.float64)
    _, uh, _, _ = o64._user_side(np.arange(helper.data_spec.user_count))
    _, vh, _ = o64._item_side(np.arange(helper.data_spec.item_count))
    want = so.evaluate_whole(uh, vh, small_data["test"][:, :2], small_data["test"][:, 1], 50)
    test_res = trainer.history[-1][4]
    # a rank flip at the top-50 boundary moves ONE user's recall by at most 1 and the mean by at most 1 / users: three flips
    assert abs(test_res["recall@50"] - want["recall"]) <= 3.0 / want["users"]
    # ... and on IDENTICAL tables (the oracle's trained tables uploaded into the engine) the product evaluator's
    # Recall / MAP / Precision@50 equal the oracle's to the north-star's 1e-5 (tie-free scores: an exact comparison)
    eng.set_tables(orc.U, orc.V)
    _, te = trainer.evaluater.run(md["model_neg_shared"], verbose=False, eval_scheme="whole")
    o32 = so.StepOracle(orc.U, orc.V, ocf, dtype=np.float64)
    _, uh, _, _ = o32._user_side(np.arange(helper.data_spec.user_count))
    _, vh, _ = o32._item_side(np.arange(helper.data_spec.item_count))
    want = so.evaluate_whole(uh, vh, small_data["test"][:, :2], small_data["test"][:, 1], 50)
    for name in ("recall", "map", "precision"):
        assert abs(te["%s@50" % name] - want[name]) <= 1e-5, (name, te, want)
    assert trainer.train_time and len(trainer.train_time) == conf.max_epoch


def test_topk_kernel_matches_reference_eval_multiple():
    """nncf_eval_topk against outputs of the reference's own eval_multiple
    (tests/golden/host_golden.npz, tools/gen_golden_host.py; tie-free scores)."""
    import ctypes
    import os
    from nncf_amd import _lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_golden.npz"))
    L = _lib.lib()
    for c in g["ev_cases"].tolist():
        pred, truth, topk = g["ev%d_pred" % c], g["ev%d_truth" % c], int(g["ev%d_topk" % c][0])
        scores = torch.as_tensor(pred.astype(np.float32)).cuda().reshape(1, -1).contiguous()
        idx = np.nonzero(truth)[0].astype(np.int32)
        ptr = torch.as_tensor(np.array([0, len(idx)], dtype=np.int32)).cuda()
        idx_t = torch.as_tensor(idx if len(idx) else np.zeros(1, dtype=np.int32)).cuda()
        met = torch.zeros(3, device="cuda")
        val = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(L.nncf_eval_topk(ctypes.c_void_p(scores.data_ptr()), 1, scores.shape[1], scores.shape[1],
                                    ctypes.c_void_p(ptr.data_ptr()), ctypes.c_void_p(idx_t.data_ptr()), topk,
                                    ctypes.c_void_p(met.data_ptr()), ctypes.c_void_p(val.data_ptr()),
                                    _lib.stream_ptr()))
        want = g["ev%d_out" % c]
        if len(idx) == 0:
            assert int(val.item()) == 0          # dropped from the mean (objectives.py:316)
            assert want.tolist() == [0., 0., 0.]
        else:
            assert int(val.item()) == 1
            np.testing.assert_allclose(met.cpu().numpy(), want, rtol=1e-6, atol=1e-7)


def _np_metrics(scores, truth_slots, topk):
    """metrics_ranking.py:14-29 with ties going to the lower slot."""
    n = len(scores)
    order = np.lexsort((np.arange(n), -scores.astype(np.float64)))[:min(topk, n)]
    hits = np.isin(order, truth_slots)
    nh = np.cumsum(hits)
    ap = float((nh / np.arange(1, len(order) + 1))[hits].sum()) / min(len(truth_slots), topk)
    return np.array([ap, hits.sum() / len(truth_slots), hits.sum() / topk])


def test_topk_selection_against_numpy():
    """The select-then-sort evaluator: random rows of every size (below a wave, beyond the old 16,384
    cap), heavy ties, rows arranged against the bucket stride (large collected sets) and one that
    overflows the LDS buffer (exact fallback), k from 1 to 1024."""
    import ctypes
    from nncf_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(5)
    cases = []
    for n, k in ((7, 50), (40, 50), (600, 50), (13600, 50), (40000, 50), (5000, 1), (5000, 1024), (3000, 700)):
        cases.append((rs.normal(size=n).astype(np.float32), k))
    cases.append((np.zeros(9000, dtype=np.float32), 50))                                  # all tied: slots 0..49
    cases.append((rs.randint(0, 5, 20000).astype(np.float32), 50))                        # 5 distinct values
    a = rs.normal(size=30000).astype(np.float32)
    a[::1024] += 100.0                                                                   # the top 30 in ONE bucket
    cases.append((a, 50))
    n = 200 * 1024
    b = rs.uniform(-1, 0, n).astype(np.float32)                                          # overflow of the LDS buffer:
    idx = np.arange(n)
    high = (idx % 1024) < 50                                                             # 50 buckets hold every large
    b[high] = 10.0 + rs.uniform(0, 1, high.sum()).astype(np.float32)                     # value, one of them only the
    b[(idx % 1024) == 49] = 5.0 + rs.uniform(0, 1, (idx % 1024 == 49).sum()).astype(np.float32)   # smallest of those
    cases.append((b, 50))
    for sc, topk in cases:
        n = len(sc)
        truth = np.sort(rs.choice(n, size=min(n, 25), replace=False)).astype(np.int32)
        top = np.lexsort((np.arange(n), -sc.astype(np.float64)))[:3]
        truth = np.unique(np.concatenate([truth, top])).astype(np.int32)                 # some hits for sure
        scores = torch.as_tensor(sc).cuda().reshape(1, -1).contiguous()
        ptr = torch.as_tensor(np.array([0, len(truth)], dtype=np.int32)).cuda()
        idx_t = torch.as_tensor(truth).cuda()
        met = torch.zeros(3, device="cuda")
        val = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(L.nncf_eval_topk(ctypes.c_void_p(scores.data_ptr()), 1, n, n, ctypes.c_void_p(ptr.data_ptr()),
                                    ctypes.c_void_p(idx_t.data_ptr()), topk, ctypes.c_void_p(met.data_ptr()),
                                    ctypes.c_void_p(val.data_ptr()), _lib.stream_ptr()))
        assert int(val.item()) == 1
        np.testing.assert_allclose(met.cpu().numpy(), _np_metrics(sc, truth, topk), rtol=1e-6, atol=1e-7,
                                   err_msg="n=%d k=%d" % (n, topk))


@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "12"))))
def test_topk_random_rows_ties_and_truth_lists(seed):
    """nncf_eval_topk over SEVERAL users at once: random row counts and lengths (1 ... 70,000 candidates; leading dimension
    above the row length), scores that are continuous, heavily tied (3 distinct values) or constant, k = 1 ... 1,024, truth
    lists that are empty (dropped from the mean, objectives.py:316), short or longer than k -- per-user AP / recall /
    precision against metrics_ranking.py:14-29 restated in numpy (ties to the lower slot)."""
    import ctypes
    from nncf_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(21000 + seed)
    rows = int(rs.choice([1, 2, 5, 17]))
    n = int(rs.choice([1, 3, 50, 64, 65, 600, 5000, 17000, 70000]))
    ld = n + int(rs.choice([0, 0, 1, 37]))
    topk = int(rs.choice([1, 5, 50, 50, 300, 1024]))
    kind = int(rs.randint(3))
    if kind == 0:
        sc = rs.normal(size=(rows, ld)).astype(np.float32)
    elif kind == 1:
        sc = rs.randint(0, 3, size=(rows, ld)).astype(np.float32)
    else:
        sc = np.full((rows, ld), 0.25, dtype=np.float32)
    ptr, idx = [0], []
    for r in range(rows):
        m = int(rs.choice([0, 1, 3, 40, 2000]))
        m = min(m, n)
        t = np.sort(rs.choice(n, size=m, replace=False)).astype(np.int32)
        idx.append(t)
        ptr.append(ptr[-1] + m)
    flat = np.concatenate(idx) if ptr[-1] else np.zeros(1, dtype=np.int32)
    scores = torch.as_tensor(sc).cuda().contiguous()
    ptr_t = torch.as_tensor(np.array(ptr, dtype=np.int32)).cuda()
    idx_t = torch.as_tensor(flat.astype(np.int32)).cuda()
    met = torch.zeros(rows * 3, device="cuda")
    val = torch.zeros(rows, dtype=torch.int32, device="cuda")
    _lib.check(L.nncf_eval_topk(ctypes.c_void_p(scores.data_ptr()), rows, ld, n, ctypes.c_void_p(ptr_t.data_ptr()),
                                ctypes.c_void_p(idx_t.data_ptr()), topk, ctypes.c_void_p(met.data_ptr()),
                                ctypes.c_void_p(val.data_ptr()), _lib.stream_ptr()))
    got, ok = met.cpu().numpy().reshape(rows, 3), val.cpu().numpy()
    for r in range(rows):
        info = (rows, n, ld, topk, kind, r, len(idx[r]))
        if len(idx[r]) == 0:
            assert ok[r] == 0, info
            continue
        assert ok[r] == 1, info
        np.testing.assert_allclose(got[r], _np_metrics(sc[r, :n], idx[r], topk), rtol=1e-6, atol=1e-7, err_msg=str(info))


def test_group_trainer_with_device_shuffle_learns():
    """conf.device_shuffle: the per-epoch stratified shuffle runs on the device (rocPRIM sorts);
    training still converges and batches are item-stratified (mean unique items per batch well
    below batch_size_p)."""
    from nncf_amd import synthetic
    from nncf_amd.conf import get_conf
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    from nncf_amd.train_group_neg_shared import Trainer
    data = synthetic.make_citeulike_like(users=400, items=300, train_links=6000, seed=21)
    pd = {"reset_after_getconf": True, "max_epoch": 3, "loss": "log-loss", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.2, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 16, "item_dim": 16, "chop_size": 4, "group_shuffling_trick": True, "device_shuffle": True,
          "interaction_bias": None, "u_reg": 1e-4}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="mf")
    helper = get_data(data, conf, rand_seed=3)
    md = get_model(conf, helper, "mf", seed=4)
    tr = Trainer(md, conf, helper)
    tr.train(None)
    costs = [h[2] for h in tr.history[1:]]
    assert all(np.isfinite(costs)) and costs[-1] < costs[0]
    assert max(tr.mean_n_u) < 40          # ~ b / chop distinct items per batch, not ~ b


@pytest.mark.parametrize("scheme", ["original", "neg_shared"])
def test_monitor_rate_all_loss_inside_the_device_epoch_loop(small_data, scheme):
    """conf.monitor_rate_all_loss (train_original.py:16-19,38-40,59-67,74-79; train_neg_shared.py:52-58): after a step, with
    probability `rate` (random.random(), one draw per step), the four loss-monitor models evaluate that step's batch
    against the parameters the step left, touching nothing.  Rate 0.3 over one epoch: the monitored steps are the
    reference's draws, the summed costs equal the oracle's, and training itself is unchanged by the cuts in the loop."""
    import random
    conf = _conf("skip-gram", scheme, epochs=1)
    conf.monitor_rate_all_loss = 0.3
    runs = []
    for monitored in (True, False):
        data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in small_data.items()}
        c2 = _conf("skip-gram", scheme, epochs=1)
        if monitored:
            c2.monitor_rate_all_loss = 0.3
        helper = get_data(data, c2, rand_seed=777)
        md = get_model(c2, helper, "mf", seed=1)
        eng = md["model"].engine
        U0, V0 = eng.U.cpu().numpy().copy(), eng.V.cpu().numpy().copy()
        mod = __import__("nncf_amd.train_%s" % scheme, fromlist=["Trainer"])
        np.random.seed(99)
        random.seed(5)
        tr = mod.Trainer(md, c2, helper)
        tr.train(None)
        runs.append((tr, eng.U.cpu().numpy().copy(), eng.V.cpu().numpy().copy()))
    (tr, U1, V1), (tr0, U2, V2) = runs
    assert np.array_equal(U1, U2) and np.array_equal(V1, V2) and tr.history[-1][2] == tr0.history[-1][2]
    # the reference loop with its monitoring, on the oracle
    from oracle.sampler_oracle import OracleSampler
    from nncf_amd.data_utils import degree_dist
    ocf = _oracle_conf(conf)
    orc = so.StepOracle(U0, V0, ocf, dtype=np.float32)
    train = small_data["train"].copy()
    b, k = conf.batch_size_p, conf.num_negatives
    smp = OracleSampler(degree_dist(train, 1, "unigram"), 1.0, 777) if scheme == "original" else None
    np.random.seed(99)
    random.seed(5)
    names = ("skip-gram", "mse", "log-loss", "max-margin")
    want = dict((n, 0.0) for n in names)
    mon_it = 0
    for epoch in range(2):
        train[:] = train[np.random.permutation(len(train))]
        for t in range(len(train) // b if epoch else 0):
            bt = train[t * b:(t + 1) * b]
            if scheme == "original":
                u, c, y = so.assemble_original(bt, smp.sample_batch(k * b), conf.loss)
                orc.step_original(u, c, y)
            else:
                u, c = bt[:, 0], bt[:, 1]
                orc.step_neg_shared(u, c)
            if random.random() < 0.3:
                mon_it += 1
                for n in names:
                    oc = _oracle_conf(conf)
                    oc.loss = n
                    o2 = so.StepOracle(orc.U, orc.V, oc, dtype=np.float32)
                    if scheme == "original":
                        want[n] += o2.step_original(u, c, np.where(np.arange(len(u)) < b, 1, -1 if n == "skip-gram" else 0), update=False)
                    else:
                        want[n] += o2.step_neg_shared(u, c, update=False)
    epoch, got_it, got = tr.monitor_history[-1]
    assert got_it == mon_it and mon_it > 10
    for n in names:
        assert abs(got[n] - want[n]) <= 1e-5 * max(1, abs(want[n])), (n, got[n], want[n])
