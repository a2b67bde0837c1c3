"""GPU: model_choice = basic_embedding (mean-pool text tower above the step kernels) against the
oracle: per-step loss, every parameter after the update, inference-mode scores."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.tolerances import assert_within_sum_bound  # noqa: E402

import torch  # noqa: E402

from oracle import step_oracle as so  # noqa: E402
from nncf_amd import synthetic  # noqa: E402
from nncf_amd.conf import get_conf  # noqa: E402
from nncf_amd.data_utils import get_data  # noqa: E402
from nncf_amd.model_framework import get_model  # noqa: E402


@pytest.fixture(scope="module")
def text_data():
    return synthetic.make_citeulike_like(users=300, items=260, train_links=5000, seed=3, content_len=24, vocab=120)


def _conf(loss, scheme, b=64, k=3, d=24, lr=0.2):
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": loss, "num_negatives": k if scheme == "original" else 0,
          "neg_loss_weight": 16, "loss_gamma": 0.1 if loss == "max-margin" else 10, "learn_rate": lr,
          "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": b, "user_dim": d, "item_dim": d,
          "word_dim": d, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.}
    return get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="basic_embedding")


def _oracle(conf, md, data, dtype=np.float32):
    core = md["model"].core
    tw, eng = core.tower, core.engine
    oc = so.OracleConf(batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
                       loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
                       learn_rate=conf.learn_rate, emb_normalization=conf.emb_normalization,
                       model_choice="basic_embedding")
    to = so.TowerOracle(data["C"], tw.W.cpu().numpy(), tw.Wd.cpu().numpy(), tw.bd.cpu().numpy(),
                        tw.gamma.cpu().numpy(), tw.beta.cpu().numpy(), conf.learn_rate, dtype=dtype)
    return so.StepOracle(eng.U.cpu().numpy(), None, oc, dtype=dtype, tower=to), to


@pytest.mark.parametrize("scheme,loss", [("neg_shared", "skip-gram"), ("neg_shared", "log-loss"),
                                         ("original", "skip-gram"), ("original", "log-loss"),
                                         ("group_neg_shared", "log-loss"), ("group_neg_shared", "max-margin")])
def test_basic_embedding_steps_match_oracle(text_data, scheme, loss):
    conf = _conf(loss, scheme)
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    core = head.core
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        if scheme == "original":
            negs = rs.randint(0, 260, b * k)
            u, c, y = so.assemble_original(bt, negs, loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        tw = core.tower
        np.testing.assert_allclose(core.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=2e-4, atol=2e-5,
                                       err_msg="%s step %d" % (name, step))
    # inference phase (BN running statistics): predict_on_batch == oracle scores
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float32)
    want = oc2.predict_neg_shared(users, np.arange(len(items)))
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("shared_path", ["fused", "tile"])
@pytest.mark.parametrize("hot_words", ["1", "0"])
@pytest.mark.parametrize("scheme,loss,opt", [("neg_shared", "skip-gram", "sgd"), ("group_neg_shared", "mse", "sgd"),
                                             ("neg_shared", "skip-gram", "adam")])
def test_every_kernel_path_of_the_shared_schemes_matches_the_oracle(text_data, monkeypatch, shared_path, hot_words, scheme, loss, opt):
    """One-launch step (step_fused.hip) vs tile + apply, frequent words' gradient rows by the Dense backward's workgroups
    (nncf_tower_set_hot_words) vs by the word kernel alone: all four combinations, 5 steps, every parameter against the
    oracle."""
    monkeypatch.setenv("NNCF_SHARED_PATH", shared_path)
    monkeypatch.setenv("NNCF_HOT_WORDS", hot_words)
    conf = _conf(loss, scheme, lr=0.02 if opt == "adam" else 0.2)
    conf.optimizer = opt
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    orc.conf.optimizer = opt
    head = md["model_neg_shared" if scheme == "neg_shared" else "model_group_neg_shared"]
    core = head.core
    b = conf.batch_size_p
    rs = np.random.RandomState(11)
    train = text_data["train"]
    for step in range(5):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1]) if scheme == "neg_shared" else orc.step_group_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        tw = core.tower
        rt, at = (5e-4, 5e-5) if opt == "adam" else (2e-5, 5e-6)     # Adam divides by sqrt(v): the existing Adam tests' budget
        np.testing.assert_allclose(core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
            if opt == "adam" and name in ("bd", "rm"):
                continue   # dL/dbd = sum dz is exactly 0 behind a BatchNorm: Adam turns its rounding noise into +-lr steps
                           # of bd, which shift the batch mean (and its running average) and nothing else
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))


@pytest.mark.parametrize("loss", ["skip-gram", "log-loss"])
def test_model_item_emb_is_the_inference_tower_of_the_given_ids(text_data, loss):
    """model_dict['model_item_emb'].predict_on_batch([cid]) (model_framework.py:174): inference-mode C_emb of
    the REQUESTED catalogue ids -- also ids beyond the per-batch block -- after a few training steps."""
    conf = _conf(loss, "neg_shared")
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    rs = np.random.RandomState(3)
    train = text_data["train"]
    for _ in range(2):
        bt = train[rs.choice(len(train), conf.batch_size_p, replace=False)]
        orc.step_neg_shared(bt[:, 0], bt[:, 1])
        md["model_neg_shared"].train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
    ids = np.array([259, 0, 7, 7, 130, 258, 64, 65, 200])       # past the block of 64, with a duplicate
    got = md["model_item_emb"].predict_on_batch([ids])
    want = to.forward(ids, train=False)
    if conf.emb_normalization:
        want, _ = so.l2_normalize(want)
    assert got.shape == (len(ids), conf.item_dim)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)
    with pytest.raises(IndexError):
        md["model_item_emb"].predict_on_batch([np.array([260])])
    users = np.array([5, 1, 1])
    got_u = md["model_user_emb"].predict_on_batch([users])
    want_u = orc.U[users]
    if conf.emb_normalization:
        want_u, _ = so.l2_normalize(want_u)
    np.testing.assert_allclose(got_u, want_u, rtol=2e-5, atol=2e-6)


def test_basic_embedding_trainer_runs_and_learns(text_data):
    conf = _conf("skip-gram", "neg_shared", lr=0.05)   # large steps kill the ReLU units (all-zero items)
    conf.max_epoch = 4
    data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in text_data.items()}
    helper = get_data(data, conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    from nncf_amd.train_neg_shared import Trainer
    np.random.seed(1)
    tr = Trainer(md, conf, helper)
    tr.train("whole")
    costs = [h[2] for h in tr.history[1:]]
    assert all(np.isfinite(costs)) and costs[-1] < costs[0] - 1e-3
    tw = md["model"].core.tower
    E = tw.forward(torch.arange(260, dtype=torch.int32, device=tw.device), train=False)
    assert float((E > 0).float().mean()) > 0.05          # the tower is alive
    assert 0 < tr.history[-1][4]["recall@50"] <= 1     # cold-start items are scored through their text


def test_epoch_runner_equals_step_by_step(text_data):
    """nncf_tower_epoch (plans batched 64 steps at a time) == nncf_tower_step per batch, bitwise."""
    for scheme, loss in (("neg_shared", "skip-gram"), ("original", "log-loss"), ("group_neg_shared", "log-loss")):
        conf = _conf(loss, scheme)
        b, k = conf.batch_size_p, conf.num_negatives
        mds = [get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2) for _ in range(2)]
        key = {"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]
        rs = np.random.RandomState(11)
        steps = 70                                           # crosses a 64-step plan chunk
        train = text_data["train"]
        bt = train[rs.choice(len(train), steps * b, replace=True)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        negs = rs.randint(0, 260, steps * b * max(k, 1)).astype(np.int32)
        h0, h1 = mds[0][key], mds[1][key]
        e = h0.engine
        uid_all, cid_all, neg_all = e.ids(bt[:, 0]), e.ids(bt[:, 1]), e.ids(negs)
        got = h0.run_epoch(uid_all, cid_all, steps, neg_all=neg_all if scheme == "original" else None).cpu().numpy()
        want = np.zeros(steps, dtype=np.float32)
        for t in range(steps):
            u, c = bt[t * b:(t + 1) * b, 0], bt[t * b:(t + 1) * b, 1]
            if scheme == "original":
                c = np.concatenate([c, negs[t * b * k:(t + 1) * b * k]])
                u = np.concatenate([u, np.repeat(u, k)])
            want[t] = h1.train_on_batch([u, c], None)
        np.testing.assert_array_equal(got, want)
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
            np.testing.assert_array_equal(getattr(h0.core.tower, name).cpu().numpy(),
                                          getattr(h1.core.tower, name).cpu().numpy(), err_msg=name)
        np.testing.assert_array_equal(h0.engine.U.cpu().numpy(), h1.engine.U.cpu().numpy())


@pytest.mark.parametrize("postings", ["1", "0"])
def test_large_block_original_matches_oracle(monkeypatch, postings):
    """More than 1024 unique items in one batch: several K-parts of the weight-gradient tiles,
    more than 64 statistics blocks.  postings: the word-table backward walks the static word -> items lists
    (nncf_tower_set_postings; default for such blocks) / scans the count matrix as the small blocks do."""
    monkeypatch.setenv("NNCF_WORD_POSTINGS", postings)
    data = synthetic.make_citeulike_like(users=400, items=3000, train_links=9000, seed=4, content_len=20, vocab=150)
    conf = _conf("skip-gram", "original", b=512, k=3, d=32, lr=0.1)
    helper = get_data(dict(data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, data)
    head = md["model"]
    rs = np.random.RandomState(3)
    for step in range(2):
        bt = data["train"][rs.choice(len(data["train"]), 512, replace=False)]
        negs = rs.randint(0, 3000, 512 * 3)
        u, c, y = so.assemble_original(bt, negs, "skip-gram")
        assert len(np.unique(c)) > 1024
        assert (getattr(head.core.tower, "post_units", None) is not None) == (postings == "1")
        want = orc.step_original(u, c, y)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        tw = head.core.tower
        np.testing.assert_allclose(head.core.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=2e-4, atol=2e-5,
                                       err_msg="%s step %d" % (name, step))


@pytest.mark.parametrize("model", ["basic_embedding", "cnn_embedding"])
def test_loss_only_evaluation_leaves_every_parameter_and_the_running_statistics_alone(model):
    """a step with update = False (the monitor models of monitor_rate_all_loss) changes NOTHING -- on every path:
    the one-launch step, the tile + apply pair, and train_scheme = original with a block of more than 1,024 rows, whose
    statistics launch / score-kernel merge used to move the BatchNorm running statistics regardless."""
    data = synthetic.make_citeulike_like(users=400, items=3000, train_links=9000, seed=4, content_len=20, vocab=150)
    for scheme, loss in (("original", "skip-gram"), ("neg_shared", "skip-gram"), ("group_neg_shared", "log-loss")):
        conf = (_conf_cnn(loss, scheme, b=512, k=3, d=32, F=20) if model == "cnn_embedding" else _conf(loss, scheme, b=512, k=3, d=32, lr=0.1))
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2)
        head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
        core, tw = head.core, head.core.tower
        rs = np.random.RandomState(3)
        bt = data["train"][rs.choice(len(data["train"]), 512, replace=False)]
        e = core.engine
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 3000, 512 * 3), loss)
            assert len(np.unique(c)) > 1024
            kind = "original"
        else:
            u, c = bt[:, 0], bt[:, 1]
            kind = "neg_shared" if scheme == "neg_shared" else "group_neg_shared"
        head.train_on_batch([u, c], [np.ones(len(u))])                      # one real step first: the statistics are not the initial ones
        names = [n for n in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv", "Wc", "bc", "g1", "b1", "rm1", "rv1") if hasattr(tw, n)]
        before = {n: getattr(tw, n).clone() for n in names}
        U0 = e.U.clone()
        l1 = float(core.step(kind, e.ids(u), e.ids(c), update=False).item())
        l2 = float(core.step(kind, e.ids(u), e.ids(c), update=False).item())
        assert np.isfinite(l1) and np.isfinite(l2)
        for n in names:
            assert torch.equal(getattr(tw, n), before[n]), (model, scheme, n)
        assert torch.equal(e.U, U0)


def test_row_dropout_matches_oracle_with_same_mask(text_data):
    """Embedding(dropout=p) row dropout: the kernels' hash mask, materialised through
    nncf_tower_dropout_scale, fed to the oracle -> same losses and parameters; masks differ per
    step and keep about (1-p) of the words."""
    conf = _conf("skip-gram", "neg_shared", lr=0.1)
    conf.word_emb_dropout_rate = 0.4
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2)
    head = md["model_neg_shared"]
    tw = head.core.tower
    assert tw.dropout == pytest.approx(0.4)
    orc, to = _oracle(conf, md, text_data)
    e = head.engine
    steps, b = 4, conf.batch_size_p
    bt = text_data["train"][np.random.RandomState(5).choice(len(text_data["train"]), steps * b, replace=False)]
    masks = [tw.dropout_scale(t).cpu().numpy() for t in range(steps)]
    assert not np.array_equal(masks[0], masks[1])
    for mk in masks:
        assert set(np.unique(mk)) <= {0.0, np.float32(1 / 0.6)}
    assert 0.45 < np.mean(np.concatenate(masks) > 0) < 0.75
    got = head.run_epoch(e.ids(bt[:, 0]), e.ids(bt[:, 1]), steps).cpu().numpy()
    for t in range(steps):
        orc.row_scale = masks[t]
        want = orc.step_neg_shared(bt[t * b:(t + 1) * b, 0], bt[t * b:(t + 1) * b, 1])
        assert abs(got[t] - want) <= 1e-5 * max(1, abs(want)), (t, got[t], want)
    for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
        np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=2e-4, atol=2e-5, err_msg=name)
    # inference ignores the mask
    items = torch.arange(0, 260, 5, dtype=torch.int32, device=tw.device)
    np.testing.assert_allclose(tw.forward(items, train=False).cpu().numpy(), to.forward(items.cpu().numpy(), train=False),
                               rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_basic_embedding_adam_matches_oracle(text_data, kind):
    conf = _conf("skip-gram", "neg_shared", lr=0.01)
    conf.optimizer = kind
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2)
    head = md["model_neg_shared"]
    tw = head.core.tower
    orc, to = _oracle(conf, md, text_data)
    orc.conf.optimizer = kind
    b = conf.batch_size_p
    rs = np.random.RandomState(9)
    for step in range(4):
        bt = text_data["train"][rs.choice(len(text_data["train"]), b, replace=False)]
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=5e-4, atol=5e-5)
        # dense_b is left out: its gradient through BatchNorm is exactly zero in real arithmetic, so
        # what Adam normalises to +-lr_t is rounding noise (in the reference as well); BatchNorm
        # removes the bias from the output, so nothing downstream depends on it.
        for name in ("W", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=5e-4, atol=5e-5,
                                       err_msg="%s step %d" % (name, step))


# ---- cnn_embedding -------------------------------------------------------------------------------

def _conf_cnn(loss, scheme, b=64, k=3, d=24, F=20, fl=3, lr=0.1, wdrop=0.0, cdrop=0.0):
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": loss, "num_negatives": k if scheme == "original" else 0,
          "neg_loss_weight": 16, "loss_gamma": 0.1 if loss == "max-margin" else 10, "learn_rate": lr,
          "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": b, "user_dim": d, "item_dim": d,
          "word_dim": 16, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": wdrop, "num_filters": [F],
          "filter_lengths": [fl], "conv_dropout_rate": cdrop}
    return get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="cnn_embedding")


def _oracle_cnn(conf, md, data, dtype=np.float32):
    core = md["model"].core
    tw, eng = core.tower, core.engine
    oc = so.OracleConf(batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
                       loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
                       learn_rate=conf.learn_rate, emb_normalization=conf.emb_normalization,
                       model_choice="basic_embedding", optimizer=conf.optimizer)
    g = lambda t: t.cpu().numpy().astype(dtype)  # noqa: E731
    to = so.CnnTowerOracle(data["C"], g(tw.W), g(tw.Wc), g(tw.bc), g(tw.g1), g(tw.b1), g(tw.Wd), g(tw.bd), g(tw.gamma),
                           g(tw.beta), conf.learn_rate, dtype=dtype)
    return so.StepOracle(g(eng.U), None, oc, dtype=dtype, tower=to), to


CNN_PARAMS = ("W", "Wc", "bc", "g1", "b1", "Wd", "bd", "gamma", "beta", "rm", "rv", "rm1", "rv1")


@pytest.mark.parametrize("scheme,loss,fl,drop", [("neg_shared", "skip-gram", 3, False), ("neg_shared", "log-loss", 3, True),
                                                 ("group_neg_shared", "log-loss", 5, False), ("original", "skip-gram", 1, True),
                                                 ("original", "skip-gram", 3, False),
                                                 ("group_neg_shared", "max-margin", 3, True)])
@pytest.mark.parametrize("pool", ["average", "max"])
@pytest.mark.parametrize("split", [0, 12])
def test_cnn_embedding_steps_match_oracle(text_data, scheme, loss, fl, drop, pool, split, monkeypatch):
    """cnn_embedding (conv + BN + relu + dropout + average | max pooling over the whole length, cnn_model.py:100-109, in
    front of Dense/BN/ReLU) against CnnTowerOracle: loss, every parameter and the running statistics after each step; the
    kernels' row-dropout and conv-dropout hash masks are materialised and fed to the oracle.  split: NNCF_CNN_SPLIT -- the conv
    backward over work units (items with more kept words cut in two, heaviest first: csrc/cnn.hip cnn_units_build)."""
    monkeypatch.setenv("NNCF_CNN_SPLIT", str(split))
    conf = _conf_cnn(loss, scheme, fl=fl, wdrop=0.3 if drop else 0.0, cdrop=0.2 if drop else 0.0)
    conf.poolings = [pool]
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "cnn_embedding", seed=2)
    orc, to = _oracle_cnn(conf, md, text_data)
    to.conv_pooling = pool
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        if drop:
            orc.row_scale = tw.dropout_scale(step).cpu().numpy()
            to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy()
        if scheme == "original":
            negs = rs.randint(0, 260, b * k)
            u, c, y = so.assemble_original(bt, negs, loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        for name in CNN_PARAMS:
            if name == "bc":   # gradient through BatchNorm is rounding noise around zero
                np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), atol=1e-5, err_msg=name)
                continue
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=3e-4, atol=3e-5,
                                       err_msg="%s step %d" % (name, step))
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float32)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=3e-4, atol=3e-5)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "12"))))
def test_cnn_tower_random_small_batches_terminate_and_match(text_data, seed):
    """cnn_embedding with random filter length / count, pooling, dropouts, scheme and loss over batches with very few unique items
    (3 ... 40 of a 64-row batch) and repeated users: every case finishes and matches the float64 CnnTowerOracle -- to the fixed
    tolerance plus four times the float32 oracle's own distance from the float64 one (what fp32 can hold in that case)."""
    rs = np.random.RandomState(800 + seed)
    scheme = str(rs.choice(["neg_shared", "group_neg_shared", "original"]))
    loss = str(rs.choice(["skip-gram", "mse", "log-loss", "max-margin"]))
    fl, F = int(rs.choice([1, 3, 5])), int(rs.choice([4, 20, 33]))
    pool, drop = str(rs.choice(["average", "max"])), bool(rs.randint(2))
    conf = _conf_cnn(loss, scheme, fl=fl, F=F, wdrop=0.3 if drop else 0.0, cdrop=0.2 if drop else 0.0)
    conf.poolings = [pool]
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "cnn_embedding", seed=2)
    orc, to = _oracle_cnn(conf, md, text_data, dtype=np.float64)
    o32, t32 = _oracle_cnn(conf, md, text_data, dtype=np.float32)   # its distance from the float64 one = what fp32 can hold in this case
    to.conv_pooling = t32.conv_pooling = pool
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    b, k = conf.batch_size_p, conf.num_negatives
    # three unique items and up: BatchNorm over TWO rows leaves +-1 and a gradient that is rounding noise times 1 / sigma (and a max
    # pooling whose winner flips on it) -- 5 of 400 such cases end 3e-5 .. 4e-4 from the float64 oracle after the second step
    n_items, n_users = int(rs.choice([3, 5, 17, 40])), int(rs.choice([1, 3, 50, 300]))
    pool_items = rs.choice(260, n_items, replace=False)
    for step in range(2):
        u = rs.randint(0, n_users, b)
        c = pool_items[rs.randint(0, n_items, b)]
        if drop:
            orc.row_scale = o32.row_scale = tw.dropout_scale(step).cpu().numpy()
            to.conv_mask = t32.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy()
        if scheme == "original":
            u, c, y = so.assemble_original(np.stack([u, c, np.ones(b, dtype=int)], 1), pool_items[rs.randint(0, n_items, b * k)], loss)
            want, w32 = orc.step_original(u, c, y), o32.step_original(u, c, y)
        elif scheme == "group_neg_shared":
            c = np.repeat(c[::4], 4)
            y = np.ones(b)
            if len(np.unique(c)) < 2:
                continue
            want, w32 = orc.step_group_neg_shared(u, c), o32.step_group_neg_shared(u, c)
        else:
            y = np.ones(b)
            if len(np.unique(c)) < 2:
                continue
            want, w32 = orc.step_neg_shared(u, c), o32.step_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        info = (scheme, loss, fl, F, pool, drop, n_items, n_users, step)
        assert np.isfinite(got), info
        # a pre-activation within rounding of 0 (or two pooled maxima within rounding of each other) makes the backward's 0 / 1 mask a
        # coin toss at fp32 -- seed 620: |pre| = 1.5e-6 in one of 1,440 places, that filter's gradient then differs by O(1e-2): stop there
        pre = to.cache["y1"]
        if np.abs(pre).min() < 2e-5:
            break
        if pool == "max":
            act = np.sort(np.maximum(pre, 0) * (to.cache["mask"] if to.cache["mask"] is not None else 1.0), axis=1)
            gap = act[:, -1] - act[:, -2]
            if ((gap > 0) & (gap < 2e-5)).any():
                break
        assert abs(got - want) <= 2e-5 * max(1, abs(want)) + 4 * abs(w32 - want), (info, got, want)
        far = lambda a32, a64: 4 * float(np.abs(np.asarray(a32, dtype=np.float64) - a64).max())   # noqa: E731
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5 + far(o32.U, orc.U), err_msg=str(info))
        for name in ("W", "Wc", "g1", "b1", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=1e-3,
                                       atol=1e-4 + far(getattr(t32, name), getattr(to, name)), err_msg=str((name,) + info))


@pytest.mark.parametrize("pool,drop", [("average", False), ("max", True)])
def test_cnn_embedding_original_block_above_512_rows_matches_oracle(pool, drop):
    """train_scheme = original with more than 512 unique items per step: the block keeps the cnn_stats / cnn_merge launches,
    each on several workgroups over slot ranges whose last arriver adds the per-workgroup rows in workgroup order (cnn.hip,
    merge_ranges).  Parity with the float64 CnnTowerOracle, and the same bits from a second model run on the same
    inputs -- whichever workgroup happens to arrive last."""
    data = synthetic.make_citeulike_like(users=300, items=1500, train_links=6000, seed=8, content_len=20, vocab=150)
    conf = _conf_cnn("skip-gram", "original", b=256, k=3, wdrop=0.3 if drop else 0.0, cdrop=0.2 if drop else 0.0)
    conf.poolings = [pool]
    runs = []
    for rep in range(2):
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "cnn_embedding", seed=2)
        head, tw = md["model"], md["model"].core.tower
        orc = to = None
        if rep == 0:
            orc, to = _oracle_cnn(conf, md, data, dtype=np.float64)   # ~9,000 terms meet in the pad word's row: the float32 oracle is
            to.conv_pooling = pool                                   # 1e-4 from the float64 one there, the kernels 4e-8
        b, k = conf.batch_size_p, conf.num_negatives
        rs = np.random.RandomState(7)
        for step in range(2):
            bt = data["train"][rs.choice(len(data["train"]), b, replace=False)]
            u, c, y = so.assemble_original(bt, rs.randint(0, 1500, b * k), "skip-gram")
            assert len(np.unique(c)) > 512
            if drop and rep == 0:
                orc.row_scale = tw.dropout_scale(step).cpu().numpy()
                to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy()
            got = head.train_on_batch([u, c], [y])
            if rep:
                continue
            want = orc.step_original(u, c, y)
            assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
            np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
            for name in CNN_PARAMS:
                if name == "bc":
                    np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), atol=1e-5, err_msg=name)
                    continue
                np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=2e-5, atol=2e-6,
                                           err_msg="%s step %d" % (name, step))
        runs.append({n: getattr(tw, n).clone() for n in CNN_PARAMS})
    for n in CNN_PARAMS:
        assert torch.equal(runs[0][n], runs[1][n]), n


def test_cnn_embedding_citeulike_tile_shape():
    """L = 200, word_dim = filters = 100: the shape whose conv forward runs the unmasked <7, 7> tile
    specialisation (two workgroups per item, 7 row tiles x 7 K groups each) and whose backward uses
    two position blocks per item."""
    data = synthetic.make_citeulike_like(users=200, items=150, train_links=3000, seed=9, content_len=200, vocab=400)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.05, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 32,
          "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.5,
          "num_filters": [100], "filter_lengths": [3], "conv_dropout_rate": 0.1}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="cnn_embedding")
    md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "cnn_embedding", seed=2)
    orc, to = _oracle_cnn(conf, md, data)
    head = md["model_neg_shared"]
    tw = head.core.tower
    rs = np.random.RandomState(1)
    for step in range(2):
        bt = data["train"][rs.choice(len(data["train"]), 32, replace=False)]
        orc.row_scale = tw.dropout_scale(step).cpu().numpy()
        to.conv_mask = tw.conv_mask(step, 32).cpu().numpy()
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        for name in CNN_PARAMS:
            if name == "bc":
                np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), atol=1e-5, err_msg=name)
                continue
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=3e-4, atol=3e-5,
                                       err_msg="%s step %d" % (name, step))


@pytest.mark.parametrize("L,wd,F,fl", [(250, 128, 128, 5), (130, 72, 88, 3)])
def test_cnn_embedding_general_tile_shapes(L, wd, F, fl):
    """Shapes outside the two fast specialisations: the masked <8, 8> forward tiles (L up to 256, word_dim up
    to 128), an 8-group / 6-group K loop in the backward, three position blocks per item, filter length 5."""
    data = synthetic.make_citeulike_like(users=120, items=90, train_links=1500, seed=13, content_len=L, vocab=300)
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "log-loss", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.05, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 32,
          "user_dim": 64, "item_dim": 64, "word_dim": wd, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.3,
          "num_filters": [F], "filter_lengths": [fl], "conv_dropout_rate": 0.1}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="cnn_embedding")
    md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "cnn_embedding", seed=2)
    orc, to = _oracle_cnn(conf, md, data, dtype=np.float64)
    head = md["model_neg_shared"]
    tw = head.core.tower
    rs = np.random.RandomState(3)
    for step in range(2):
        bt = data["train"][rs.choice(len(data["train"]), 32, replace=False)]
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        to.conv_mask = tw.conv_mask(step, 32).cpu().numpy().astype(np.float64)
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        for name in CNN_PARAMS:
            if name == "bc":
                np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), atol=1e-5, err_msg=name)
                continue
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=3e-4, atol=3e-5,
                                       err_msg="%s step %d" % (name, step))


@pytest.mark.parametrize("model", ["basic_embedding", "cnn_embedding"])
def test_dims_not_multiple_of_4_are_padded_exactly(text_data, model):
    """The reference's default widths are 50: the wrappers zero-pad every width to a multiple of 4.
    The padded model against the oracle run at the CONFIGURED widths (parameters sliced to the
    logical dims): same losses, same logical parameters; pad rows / columns stay exactly zero."""
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "log-loss", "num_negatives": 0, "neg_loss_weight": 16,
          "loss_gamma": 10, "learn_rate": 0.1, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 64,
          "user_dim": 18, "item_dim": 18, "word_dim": 10, "chop_size": 4, "u_reg": 1e-4, "word_emb_dropout_rate": 0.,
          "num_filters": [14], "filter_lengths": [3], "conv_dropout_rate": 0.}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice=model)
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), model, seed=2)
    head = md["model_neg_shared"]
    tw, eng = head.core.tower, head.engine
    assert (eng.dim, eng.dim_p, tw.wd_l, tw.wd, tw.d) == (18, 20, 10, 12, 20)
    g = lambda t: t.cpu().numpy()  # noqa: E731
    oc = so.OracleConf(batch_size_p=64, num_negatives=0, loss="log-loss", loss_gamma=10, neg_loss_weight=16, u_reg=1e-4,
                       learn_rate=0.1, emb_normalization=conf.emb_normalization, model_choice="basic_embedding")
    d, wd, hd = 18, 10, tw.hd_l
    if model == "cnn_embedding":
        to = so.CnnTowerOracle(text_data["C"], g(tw.W)[:, :wd], g(tw.Wc)[:, :wd, :hd], g(tw.bc)[:hd], g(tw.g1)[:hd], g(tw.b1)[:hd],
                               g(tw.Wd)[:hd, :d], g(tw.bd)[:d], g(tw.gamma)[:d], g(tw.beta)[:d], 0.1, dtype=np.float32)
        names = {"W": (slice(None), slice(0, wd)), "Wc": (slice(None), slice(0, wd), slice(0, hd)), "g1": (slice(0, hd),),
                 "b1": (slice(0, hd),), "Wd": (slice(0, hd), slice(0, d)), "gamma": (slice(0, d),), "beta": (slice(0, d),)}
    else:
        to = so.TowerOracle(text_data["C"], g(tw.W)[:, :wd], g(tw.Wd)[:hd, :d], g(tw.bd)[:d], g(tw.gamma)[:d], g(tw.beta)[:d],
                            0.1, dtype=np.float32)
        names = {"W": (slice(None), slice(0, wd)), "Wd": (slice(0, hd), slice(0, d)), "gamma": (slice(0, d),), "beta": (slice(0, d),)}
    orc = so.StepOracle(g(eng.U)[:, :d], None, oc, dtype=np.float32, tower=to)
    rs = np.random.RandomState(3)
    for step in range(3):
        bt = text_data["train"][rs.choice(len(text_data["train"]), 64, replace=False)]
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
    np.testing.assert_allclose(g(eng.U)[:, :d], orc.U, rtol=2e-5, atol=5e-6)
    assert np.all(g(eng.U)[:, d:] == 0)
    for name, sl in names.items():
        full = g(getattr(tw, name))
        np.testing.assert_allclose(full[sl], getattr(to, name), rtol=3e-4, atol=3e-5, err_msg=name)
        mask = np.ones(full.shape, dtype=bool)
        mask[sl] = False
        if name not in ("gamma", "g1"):            # BatchNorm scales of pad channels keep their initial 1
            assert np.all(full[mask] == 0), name
    E = head.core.eval_engine().V.cpu().numpy()
    assert np.all(E[:, d:] == 0)


def test_tower_edge_cases(text_data):
    """A batch whose items are all the same item (n_u = 1: BatchNorm over one row), an item made of
    pad tokens only, inference on zero rows."""
    data = dict(text_data)
    C = data["C"].copy()
    C[5, :] = 0                                                            # all-pad item
    data["C"] = C
    for model in ("basic_embedding", "cnn_embedding"):
        conf = _conf_cnn("skip-gram", "neg_shared") if model == "cnn_embedding" else _conf("skip-gram", "neg_shared")
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2)
        orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, data)
        head = md["model_neg_shared"]
        b = conf.batch_size_p
        rs = np.random.RandomState(0)
        for cid in (np.full(b, 5), np.full(b, 17), np.r_[np.full(b // 2, 5), rs.randint(0, 260, b // 2)]):
            uid = rs.randint(0, 300, b)
            want = orc.step_neg_shared(uid, cid)
            got = head.train_on_batch([uid, cid], None)
            assert np.isfinite(got) and abs(got - want) <= 5e-5 * max(1, abs(want)), (model, got, want)
        tw = head.core.tower
        # (BatchNorm over ONE row puts every pre-activation exactly on the ReLU threshold: which side
        # rounding noise falls on is not comparable between implementations -- losses are, and
        # every parameter must stay finite)
        for name in ("W", "Wd", "gamma", "beta", "rm", "rv"):
            assert np.all(np.isfinite(getattr(tw, name).cpu().numpy())), name
        empty = torch.zeros(0, dtype=torch.int32, device=tw.device)
        assert tuple(tw.forward(empty, train=False).shape) == (0, tw.d)


@pytest.mark.parametrize("scheme,loss", [("neg_shared", "skip-gram"), ("group_neg_shared", "log-loss"), ("original", "mse")])
def test_interaction_bias_with_text_tower(text_data, scheme, loss):
    """interaction_bias = 'both' above a text tower: the item bias is indexed by catalogue id while the
    step kernels see tf.unique ranks of a per-batch block."""
    conf = _conf(loss, scheme)
    conf.interaction_bias = "both"
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    orc.conf.interaction_bias = "both"
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    eng = head.engine
    rs = np.random.RandomState(7)
    eng.bias["ub"].copy_(torch.as_tensor(rs.randn(300).astype(np.float32) * 0.2))
    eng.bias["cb"].copy_(torch.as_tensor(rs.randn(260).astype(np.float32) * 0.2))
    orc.ub, orc._want_cb = eng.bias["ub"].cpu().numpy().copy(), True
    orc.cb = eng.bias["cb"].cpu().numpy().copy()
    b, k = conf.batch_size_p, conf.num_negatives
    train = text_data["train"]
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.bias["ub"].cpu().numpy(), orc.ub, rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), orc.cb, rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(head.core.tower.W.cpu().numpy(), to.W, rtol=2e-4, atol=2e-5)
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float32)
    oc2.ub, oc2.cb = orc.ub, orc.cb[items]
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module")
def citeulike_data():
    return synthetic.make_citeulike_like(content_len=200, vocab=8001)     # 16K users x 17K items x 200K links + text


@pytest.mark.parametrize("model,scheme", [("basic_embedding", "neg_shared"), ("basic_embedding", "group_neg_shared"),
                                          ("basic_embedding", "original"), ("cnn_embedding", "group_neg_shared")])
def test_text_towers_at_baseline_shape(citeulike_data, model, scheme):
    """BASELINE configs[0..2] and [4] at their full shape (b = 512, d = 100, L = 200, k = 10, word-row
    dropout 0.5, conv dropout 0.1) for two steps against the oracle: loss, user table, tower parameters."""
    data = citeulike_data
    loss = "log-loss" if scheme == "group_neg_shared" else "skip-gram"
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": loss, "num_negatives": 10 if scheme == "original" else 0,
          "neg_loss_weight": 128, "loss_gamma": 10, "learn_rate": 0.01, "neg_dist": "unigram", "neg_sampling_power": 1,
          "batch_size_p": 512, "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 5, "u_reg": 1e-6,
          "word_emb_dropout_rate": 0.5, "num_filters": [100], "filter_lengths": [3], "conv_dropout_rate": 0.1}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice=model)
    md = get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2)
    # float64 oracle: at this shape a float32 restatement is the noisier side (the pad word's row is a sum over ~75
    # occurrences in each of up to 5,632 items)
    orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, data, dtype=np.float64)
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    init = {n: getattr(tw, n).cpu().numpy().copy() for n in ("W", "Wc", "g1", "b1", "Wd", "gamma", "beta") if hasattr(tw, n)}
    b, k = 512, conf.num_negatives
    rs = np.random.RandomState(7)
    train = data["train"]
    n_items = data["C"].shape[0]
    for step in range(2):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy()
        if model == "cnn_embedding":
            to.conv_mask = tw.conv_mask(step, b).cpu().numpy()
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, n_items, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)        # north-star: 1e-5
    np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=1e-5, atol=2e-6)
    names = ("W", "Wc", "g1", "b1", "Wd", "gamma", "beta") if model == "cnn_embedding" else ("W", "Wd", "gamma", "beta")
    # Parameters against the float64 oracle with bounds that follow from the number of summed terms (tests/tolerances.py),
    # not flat constants: a word row is an fp32 sum over every (item, occurrence) of the word in the block -- up to
    # rows x L terms for the pad id --, a Dense / BatchNorm / conv entry a sum over the block's rows (x L positions for the conv)
    rows = b * (1 + k) if scheme == "original" else b
    L = data["C"].shape[1]
    terms = {"W": rows * L, "Wc": rows * L, "g1": rows * L, "b1": rows * L, "Wd": rows, "gamma": rows, "beta": rows}
    for name in names:
        assert_within_sum_bound(getattr(tw, name).cpu().numpy(), getattr(to, name), init[name], terms[name], 2, name)


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_cnn_embedding_adam_matches_oracle(text_data, kind):
    conf = _conf_cnn("skip-gram", "neg_shared", lr=0.01)
    conf.optimizer = kind
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "cnn_embedding", seed=2)
    head = md["model_neg_shared"]
    tw = head.core.tower
    orc, to = _oracle_cnn(conf, md, text_data)
    b = conf.batch_size_p
    rs = np.random.RandomState(9)
    for step in range(4):
        bt = text_data["train"][rs.choice(len(text_data["train"]), b, replace=False)]
        want = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        got = head.train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=5e-4, atol=5e-5)
        # conv / dense biases sit in front of a BatchNorm: their gradients are rounding noise that Adam
        # normalises to +-lr_t (see test_basic_embedding_adam_matches_oracle)
        for name in ("W", "Wc", "g1", "b1", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=1e-3, atol=1e-4,
                                       err_msg="%s step %d" % (name, step))


@pytest.mark.parametrize("model,loss", [("basic_embedding", "skip-gram"), ("basic_embedding", "mse"),
                                        ("cnn_embedding", "skip-gram"), ("cnn_embedding", "mse")])
def test_rowwise_batches_above_a_text_tower(text_data, model, loss):
    """presample / group_sample / reverse feed model.train_on_batch b(1+k) (user, item, label) rows in an
    arbitrary order (the demo scripts run them with cnn_embedding): the general row-wise kernels with the
    item rows coming from the tower == the oracle's row-wise step; the epoch runner == step by step."""
    conf = (_conf_cnn if model == "cnn_embedding" else _conf)(loss, "original")
    helper = get_data(dict(text_data), conf, rand_seed=5)
    mds = [get_model(conf, helper, model, seed=2) for _ in range(2)]
    md = mds[0]
    # CNN tower: float64 oracle.  With ~250 items x 24 positions behind the conv BatchNorm, the numpy fp32
    # restatement of its backward (N da - sum da - xhat sum da xhat) loses 5e-4 on the word rows to
    # cancellation; the kernels agree with the float64 evaluation to 1e-7.
    orc, to = _oracle_cnn(conf, md, text_data, dtype=np.float64) if model == "cnn_embedding" else _oracle(conf, md, text_data)
    head = md["model"]
    b, k = conf.batch_size_p, conf.num_negatives
    n = b * (1 + k)
    rs = np.random.RandomState(17)
    train = text_data["train"]
    rows = []
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
        if step == 1:      # reverse: the negatives are sampled USERS of the positive's item
            u = u.copy()
            u[b:] = rs.randint(0, 300, b * k)
            c = np.concatenate([c[:b], np.repeat(c[:b], k)])
        perm = rs.permutation(n)
        u, c, y = u[perm], c[perm], y[perm]
        rows.append((u, c, y))
        want = orc.step_original(u, c, y)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        names = CNN_PARAMS if model == "cnn_embedding" else ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
        for name in names:
            np.testing.assert_allclose(getattr(head.core.tower, name).cpu().numpy(), getattr(to, name), rtol=2e-4, atol=2e-5,
                                       err_msg="%s step %d" % (name, step))
    # the epoch runner on a fresh model: same losses, same parameters (bitwise)
    h1 = mds[1]["model"]
    e = h1.engine
    cat = lambda i: e.ids(np.concatenate([r[i] for r in rows]))  # noqa: E731
    losses = h1.run_epoch_rowwise(cat(0), cat(1), cat(2), len(rows)).cpu().numpy()
    assert torch.equal(h1.engine.U, head.engine.U) and torch.equal(h1.core.tower.W, head.core.tower.W)
    assert losses.shape == (3,) and np.all(np.isfinite(losses))


@pytest.mark.parametrize("model", ["basic_embedding", "cnn_embedding"])
def test_sampled_neg_shared_above_a_text_tower(text_data, model):
    """model_sampled_neg_shared with a text tower: b positives + k negatives shared by every positive."""
    conf = (_conf_cnn if model == "cnn_embedding" else _conf)("skip-gram", "original")
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, model, seed=2)
    orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, text_data)
    head = md["model_sampled_neg_shared"]
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(23)
    train = text_data["train"]
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        negs = rs.randint(0, 260, k)
        u = np.concatenate([bt[:, 0], np.zeros(k, dtype=np.int64)])
        c = np.concatenate([bt[:, 1], negs])
        want = orc.step_sampled_neg_shared(u, c)
        got = head.train_on_batch([u, c])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(head.core.tower.W.cpu().numpy(), to.W, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(head.core.tower.Wd.cpu().numpy(), to.Wd, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_rowwise_batches_above_the_cnn_tower_under_adam(text_data, kind):
    """The demo scripts' presample / group_sample runs: cnn_embedding, row-wise batches, Adam."""
    conf = _conf_cnn("skip-gram", "original", lr=0.01)
    conf.optimizer = kind
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "cnn_embedding", seed=2)
    head = md["model"]
    tw = head.core.tower
    orc, to = _oracle_cnn(conf, md, text_data, dtype=np.float64)
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(31)
    for step in range(3):
        bt = text_data["train"][rs.choice(len(text_data["train"]), b, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), "skip-gram")
        perm = rs.permutation(b * (1 + k))
        u, c, y = u[perm], c[perm], y[perm]
        want = orc.step_original(u, c, y)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=5e-4, atol=5e-5)
        for name in ("W", "Wc", "g1", "b1", "Wd", "gamma", "beta"):   # bc / bd: see test_cnn_embedding_adam_matches_oracle
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=1e-3, atol=1e-4,
                                       err_msg="%s step %d" % (name, step))


def test_rowwise_tower_with_normalisation_and_bias(text_data):
    """Row-wise batches above the mean-pool tower with emb_normalization (both sides l2-normalised) and
    interaction_bias = 'both' (item bias by catalogue id, block rows by tf.unique rank)."""
    conf = _conf("skip-gram", "original")
    conf.emb_normalization = True
    conf.interaction_bias = "both"
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    orc.conf.interaction_bias = "both"
    head = md["model"]
    eng = head.engine
    rs = np.random.RandomState(41)
    eng.bias["ub"].copy_(torch.as_tensor(rs.randn(300).astype(np.float32) * 0.2))
    eng.bias["cb"].copy_(torch.as_tensor(rs.randn(260).astype(np.float32) * 0.2))
    orc.ub, orc._want_cb = eng.bias["ub"].cpu().numpy().copy(), True
    orc.cb = eng.bias["cb"].cpu().numpy().copy()
    b, k = conf.batch_size_p, conf.num_negatives
    for step in range(3):
        bt = text_data["train"][rs.choice(len(text_data["train"]), b, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), "skip-gram")
        perm = rs.permutation(b * (1 + k))
        u, c, y = u[perm], c[perm], y[perm]
        want = orc.step_original(u, c, y)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(eng.U.cpu().numpy(), orc.U, rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(eng.bias["ub"].cpu().numpy(), orc.ub, rtol=2e-5, atol=5e-6)
        np.testing.assert_allclose(eng.bias["cb"].cpu().numpy(), orc.cb, rtol=2e-5, atol=5e-6)
        for name in ("W", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(head.core.tower, name).cpu().numpy(), getattr(to, name), rtol=2e-4, atol=2e-5,
                                       err_msg="%s step %d" % (name, step))
    # the loss monitor head (SGD(0.), model_all_loss) scores a row-wise batch without touching anything
    mon = md["model_all_loss"]["skip-gram"]
    U0 = eng.U.clone()
    lm = mon.train_on_batch([u, c], [y])
    assert np.isfinite(lm) and torch.equal(eng.U, U0)


@pytest.mark.parametrize("model", ["mf", "basic_embedding"])
def test_row_order_does_not_matter_at_baseline_shape(citeulike_data, model):
    """Size-independent property at BASELINE's full shape (b = 512, k = 10, d = 100, 16K x 17K): the b(1+k)
    (user, item, label) rows of an `original` batch give the same loss and the same updated parameters
    whether they arrive positives-first (per-positive kernels) or in any other order (row-wise kernels)."""
    data = citeulike_data
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 10, "neg_loss_weight": 128,
          "loss_gamma": 10, "learn_rate": 0.01, "neg_dist": "unigram", "neg_sampling_power": 1, "batch_size_p": 512,
          "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 5, "u_reg": 1e-6, "word_emb_dropout_rate": 0.,
          "interaction_bias": None}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice=model)
    mds = [get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2) for _ in range(2)]
    b, k = 512, 10
    rs = np.random.RandomState(3)
    n_items = data["C"].shape[0]
    for step in range(2):
        bt = data["train"][rs.choice(len(data["train"]), b, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, n_items, b * k), "skip-gram")
        perm = rs.permutation(b * (1 + k))
        l0 = mds[0]["model"].train_on_batch([u, c], [y])
        l1 = mds[1]["model"].train_on_batch([u[perm], c[perm]], [y[perm]])
        assert abs(l0 - l1) <= 1e-5 * max(1, abs(l0)), (step, l0, l1)
    e0, e1 = mds[0]["model"].engine, mds[1]["model"].engine
    np.testing.assert_allclose(e0.U.cpu().numpy(), e1.U.cpu().numpy(), rtol=1e-5, atol=1e-6)
    if model == "mf":
        np.testing.assert_allclose(e0.V.cpu().numpy(), e1.V.cpu().numpy(), rtol=1e-5, atol=1e-6)
    else:
        t0, t1 = mds[0]["model"].core.tower, mds[1]["model"].core.tower
        for name in ("W", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(t0, name).cpu().numpy(), getattr(t1, name).cpu().numpy(), rtol=1e-4, atol=2e-6,
                                       err_msg=name)


@pytest.mark.parametrize("model,scheme", [("basic_embedding", "neg_shared"), ("cnn_embedding", "group_neg_shared"),
                                          ("cnn_embedding", "original")])
def test_steps_are_bitwise_reproducible_at_baseline_shape(citeulike_data, model, scheme):
    """No float atomics, fixed summation orders, and a conv-backward piece queue whose pieces do not depend on
    which wave computes them: two fresh models fed the same batches end up bit-identical (BASELINE shape)."""
    data = citeulike_data
    loss = "log-loss" if scheme == "group_neg_shared" else "skip-gram"
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": loss, "num_negatives": 10 if scheme == "original" else 0,
          "neg_loss_weight": 128, "loss_gamma": 10, "learn_rate": 0.01, "neg_dist": "unigram", "neg_sampling_power": 1,
          "batch_size_p": 512, "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 5, "u_reg": 1e-6,
          "word_emb_dropout_rate": 0.5, "num_filters": [100], "filter_lengths": [3], "conv_dropout_rate": 0.1}
    conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice=model)
    key = {"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]
    b, k = 512, conf.num_negatives
    n_items = data["C"].shape[0]
    outs = []
    for run in range(2):
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2)
        head = md[key]
        rs = np.random.RandomState(11)
        losses = []
        for step in range(3):
            bt = data["train"][rs.choice(len(data["train"]), b, replace=False)]
            if scheme == "group_neg_shared":
                bt[:, 1] = np.repeat(bt[::4, 1], 4)
            if scheme == "original":
                u, c, y = so.assemble_original(bt, rs.randint(0, n_items, b * k), loss)
            else:
                u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            losses.append(head.train_on_batch([u, c], [y]))
        tw = head.core.tower
        names = ("W", "Wd", "bd", "gamma", "beta", "rm", "rv") + (("Wc", "bc", "g1", "b1") if model == "cnn_embedding" else ())
        outs.append((losses, head.engine.U.clone(), {n: getattr(tw, n).clone() for n in names}))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    for n, v in outs[0][2].items():
        assert torch.equal(v, outs[1][2][n]), n


def _bank_oracle(conf, md, data, dtype=np.float32):
    """CnnBankTowerOracle at the CONFIGURED widths from the (zero-padded) device parameters"""
    core = md["model"].core
    tw, eng = core.tower, core.engine
    oc = so.OracleConf(batch_size_p=conf.batch_size_p, num_negatives=conf.num_negatives, loss=conf.loss,
                       loss_gamma=conf.loss_gamma, neg_loss_weight=conf.neg_loss_weight, u_reg=conf.u_reg,
                       learn_rate=conf.learn_rate, emb_normalization=conf.emb_normalization,
                       model_choice="basic_embedding", optimizer=conf.optimizer)
    g = lambda t: t.cpu().numpy().astype(dtype)  # noqa: E731
    wd, F, d = tw.wd_l, tw.F_l, tw.d_l
    rows = tw.dense_rows().numpy()
    to = so.CnnBankTowerOracle(
        data["C"], g(tw.W)[:, :wd], [g(tw.bank_Wc(b))[:, :wd, :F] for b in range(tw.nb)], [g(tw.bc)[b, :F] for b in range(tw.nb)],
        [g(tw.g1)[b, :F] for b in range(tw.nb)], [g(tw.b1)[b, :F] for b in range(tw.nb)], g(tw.Wd)[rows][:, :d], g(tw.bd)[:d],
        g(tw.gamma)[:d], g(tw.beta)[:d], conf.learn_rate, dtype=dtype)
    return so.StepOracle(g(eng.U)[:, :eng.dim], None, oc, dtype=dtype, tower=to), to


def _check_bank(tw, to, step):
    g = lambda t: t.cpu().numpy()  # noqa: E731
    wd, F, d = tw.wd_l, tw.F_l, tw.d_l
    rows = tw.dense_rows().numpy()
    tol = dict(rtol=3e-4, atol=3e-5)
    # the pad word's row sums three banks x ~15 positions x every item of the block: float32 summation order shows
    np.testing.assert_allclose(g(tw.W)[:, :wd], to.W, err_msg="W step %d" % step, rtol=3e-4, atol=1e-4)
    np.testing.assert_allclose(g(tw.Wd)[rows][:, :d], to.Wd, err_msg="Wd step %d" % step, **tol)
    for name in ("bd", "gamma", "beta", "rm", "rv"):
        np.testing.assert_allclose(g(getattr(tw, name))[:d], getattr(to, name), err_msg=name, **tol)
    for b in range(tw.nb):
        np.testing.assert_allclose(g(tw.bank_Wc(b))[:, :wd, :F], to.Wc[b], err_msg="Wc[%d] step %d" % (b, step), **tol)
        np.testing.assert_allclose(g(tw.bc)[b, :F], to.bc[b], atol=1e-5, err_msg="bc[%d]" % b)   # rounding noise around zero
        for name in ("g1", "b1", "rm1", "rv1"):
            np.testing.assert_allclose(g(getattr(tw, name))[b, :F], getattr(to, name)[b], err_msg="%s[%d] step %d" % (name, b, step), **tol)
    # pad rows / columns stay exactly zero: the padded model IS the configured one
    pad_rows = np.setdiff1d(np.arange(tw.hd), rows)
    assert np.all(g(tw.Wd)[pad_rows] == 0) and np.all(g(tw.Wd)[:, d:] == 0) and np.all(g(tw.W)[:, wd:] == 0)
    assert np.all(g(tw.Wc)[:, wd:, :] == 0) and np.all(g(tw.Wc)[:, :, F:] == 0)


@pytest.mark.parametrize("scheme,loss,drop,opt", [("neg_shared", "skip-gram", False, "sgd"), ("group_neg_shared", "log-loss", True, "sgd"),
                                                  ("original", "skip-gram", True, "sgd"), ("neg_shared", "log-loss", True, "adam")])
@pytest.mark.parametrize("pool", ["average", "max"])
def test_cnn_embedding_default_preset_filter_bank(text_data, scheme, loss, drop, opt, pool):
    """cnn_embedding with the reference's `default` preset front (cnn_embedding_conf.py:30-50): widths 50 (zero-padded
    to 52 on the device), filter_lengths [[1, 3, 5]] x 50 filters merged by concatenation (cnn_model.py:63-81), one
    BatchNorm over the 150 channels, relu, Dropout, average pooling, Dense(150 -> 50).  Every step against
    CnnBankTowerOracle at the configured widths: loss, all parameters of the three banks, running statistics, the pad
    rows / columns exactly zero, and the inference-mode scores."""
    pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": loss, "num_negatives": 3 if scheme == "original" else 0,
          "neg_loss_weight": 16, "loss_gamma": 10, "learn_rate": 0.1 if opt == "sgd" else 0.01, "neg_dist": "unigram",
          "neg_sampling_power": 1, "batch_size_p": 64, "user_dim": 50, "item_dim": 50, "word_dim": 50, "chop_size": 4,
          "u_reg": 1e-4, "word_emb_dropout_rate": 0.3 if drop else 0., "num_filters": [50], "filter_lengths": [[1, 3, 5]],
          "conv_dropout_rate": 0.3 if drop else 0., "poolings": [pool]}
    conf = get_conf("synthetic", "best", dict({"optimizer": opt}, **pd), model_choice="cnn_embedding")
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "cnn_embedding", seed=2)
    orc, to = _bank_oracle(conf, md, text_data, dtype=np.float64)   # float64: three banks' rounding would otherwise be the oracle's too
    to.conv_pooling = pool
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    assert (tw.nb, tw.filter_lens, tw.F_l, tw.F, tw.hd_l, tw.hd, tw.wd, tw.d) == (3, [1, 3, 5], 50, 52, 150, 156, 52, 52)
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        if drop:
            orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
            m = tw.conv_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
            to.conv_mask = [m[g][:, :, :tw.F_l] for g in range(tw.nb)]
        if scheme == "original":
            negs = rs.randint(0, 260, b * k)
            u, c, y = so.assemble_original(bt, negs, loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        if opt == "sgd":
            np.testing.assert_allclose(head.engine.U.cpu().numpy()[:, :50], orc.U, rtol=1e-4, atol=2e-5)
            _check_bank(tw, to, step)
        else:
            # conv / dense biases sit in front of a BatchNorm: their gradients are rounding noise that Adam normalises
            # to +-lr_t (test_cnn_embedding_adam_matches_oracle) -- everything else, per bank
            g = lambda t: t.cpu().numpy()  # noqa: E731

            def close(got_, want_, what):
                # lr_t ~ 0.01: an element whose gradient is rounding noise takes a step of up to +-lr_t either way; all but a
                # handful of elements agree tightly, none is further off than two such steps per step taken
                err = np.abs(got_ - want_)
                bad = err > 1e-4 + 1e-3 * np.abs(want_)
                assert bad.mean() <= 2e-3 and err.max() <= 0.02 * (step + 1), (what, step, bad.mean(), err.max())
            np.testing.assert_allclose(g(head.engine.U)[:, :50], orc.U, rtol=5e-4, atol=5e-5)
            close(g(tw.W)[:, :50], to.W, "W")
            close(g(tw.Wd)[tw.dense_rows().numpy()][:, :50], to.Wd, "Wd")
            for bk in range(tw.nb):
                close(g(tw.bank_Wc(bk))[:, :50, :50], to.Wc[bk], "Wc[%d]" % bk)
                for name in ("g1", "b1"):
                    if pool == "max" and name == "b1":
                        continue   # the maximum always sits at a kept, live position: m[p*] = 1 / keep for every item, so b1 shifts
                                   # h uniformly over the batch and the Dense layer's BatchNorm removes it -- dL/db1 = 0 but for
                                   # rounding noise, which Adam normalises (as it does for bc and bd)
                    close(g(getattr(tw, name))[bk, :50], getattr(to, name)[bk], name)
    if opt != "sgd":
        return    # inference uses the running statistics: the noise-driven bias steps are not normalised away there
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float64)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=3e-4, atol=3e-5)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NNCF_SOAK_SEEDS", "10"))))
def test_text_tower_random_small_batches_terminate_and_match(text_data, seed):
    """basic_embedding through the one-launch step with very few unique items (1 ... 40 of a 64-row batch: most item
    strips empty, BatchNorm over a handful of rows) and with users repeated: every case finishes (the in-launch
    hand-shakes count arrivals from the plans) and matches the oracle."""
    rs = np.random.RandomState(500 + seed)
    scheme = ["neg_shared", "group_neg_shared"][seed % 2]
    loss = ["skip-gram", "mse"][(seed // 2) % 2]
    conf = _conf(loss, scheme)
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data)
    head = md["model_neg_shared" if scheme == "neg_shared" else "model_group_neg_shared"]
    b = conf.batch_size_p
    n_items = int(rs.choice([2, 3, 5, 17, 40]))
    n_users = int(rs.choice([1, 3, 50, 300]))
    pool_items = rs.choice(260, n_items, replace=False)
    for step in range(2):
        u = rs.randint(0, n_users, b)
        c = pool_items[rs.randint(0, n_items, b)]
        if scheme == "group_neg_shared":
            c = np.repeat(c[::4], 4)
            if len(np.unique(c)) < 2:
                continue
            want = orc.step_group_neg_shared(u, c)
        else:
            want = orc.step_neg_shared(u, c)
        got = head.train_on_batch([u, c], [np.ones(b)])
        assert abs(got - want) <= 5e-5 * max(1, abs(want)), (step, got, want)
        tw = head.core.tower
        np.testing.assert_allclose(head.engine.U.cpu().numpy(), orc.U, rtol=1e-4, atol=2e-5)
        for name in ("W", "Wd", "gamma", "beta"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=1e-3, atol=1e-4, err_msg=name)


@pytest.mark.parametrize("pooling,content_id", [("max", False), ("average", True), ("max", True)])
@pytest.mark.parametrize("scheme,loss,opt", [("neg_shared", "skip-gram", "sgd"), ("original", "skip-gram", "sgd"),
                                             ("group_neg_shared", "log-loss", "sgd"), ("neg_shared", "skip-gram", "adam")])
def test_pooling_max_and_use_content_id_match_oracle(text_data, pooling, content_id, scheme, loss, opt):
    """conf.pooling = 'max' (mean_pool.py:70-72) and conf.use_content_id with v_reg > 0 (mean_pool.py:102-108): three
    steps with word-row dropout (masks materialised for the oracle) -- loss, user table, every tower parameter, the
    Emb_Cid table -- then inference-mode scores and model_item_emb."""
    conf = _conf(loss, scheme, lr=0.02 if opt == "adam" else 0.2)
    conf.optimizer = opt
    conf.pooling = pooling
    conf.use_content_id = content_id
    conf.v_reg = 0.05 if content_id else 0.0
    conf.word_emb_dropout_rate = 0.3
    helper = get_data(dict(text_data), conf, rand_seed=5)
    md = get_model(conf, helper, "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data, dtype=np.float64)
    orc.conf.optimizer = opt
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    core = head.core
    tw = core.tower
    to.pooling = pooling
    if content_id:
        to.Cid, to.v_reg = tw.Cid.cpu().numpy().astype(np.float64), 0.05
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    rt, at = (5e-4, 5e-5) if opt == "adam" else (2e-5, 5e-6)
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv") + (("Cid",) if content_id else ()):
            if opt == "adam" and name in ("bd", "rm"):
                continue
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))
    if opt == "adam":
        return    # bd (and with it the running mean) carries Adam's +-lr rounding steps: inference is checked under SGD
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float64)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=2e-4, atol=2e-5)
    emb = md["model_item_emb"].predict_on_batch([items])
    if conf.emb_normalization:
        E = E / np.sqrt(np.maximum((E * E).sum(-1, keepdims=True), 1e-12))
    np.testing.assert_allclose(emb, E, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("model,scheme,loss,opt,big", [
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", False),       # one-launch step: the term joins the deferred loss
    ("basic_embedding", "original", "skip-gram", "sgd", False),
    ("basic_embedding", "group_neg_shared", "log-loss", "adam", False),
    ("basic_embedding", "original", "skip-gram", "sparse_adam", True),   # 2,048-row block: postings walk + hot words
    ("cnn_embedding", "neg_shared", "skip-gram", "sgd", False),
    ("cnn_embedding", "original", "log-loss", "sgd", False),
])
def test_c_reg_matches_oracle(text_data, model, scheme, loss, opt, big):
    """conf.c_reg (mean_pool.py:56-59, cnn_model.py:45-48: activity_l2(c_reg) on the word Embedding's output; L1L2Act,
    utilities.py:122-135): three steps with word-row dropout against the float64 oracle -- loss (with the regulariser), user
    table, every tower parameter -- a loss-only step in between (the term belongs to the loss, the counts must be left clean),
    and the same steps with c_reg = 0 must differ (the term is not ignored)."""
    c_reg = 0.05
    data = text_data
    if big:
        data = synthetic.make_citeulike_like(users=400, items=3000, train_links=9000, seed=4, content_len=24, vocab=300)
    n_items = len(data["C"])

    def make(creg):
        if model == "cnn_embedding":
            conf = _conf_cnn(loss, scheme, wdrop=0.3, cdrop=0.2)
        else:
            conf = _conf(loss, scheme, lr=0.02 if opt != "sgd" else 0.2, **(dict(b=512, k=3, d=32) if big else {}))
            conf.word_emb_dropout_rate = 0.3
        conf.optimizer = opt
        conf.c_reg = creg
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), model, seed=2)
        return conf, md

    conf, md = make(c_reg)
    _, md0 = make(0.0)
    orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, data, dtype=np.float64)
    orc.conf.optimizer = opt
    to.c_reg = c_reg
    key = {"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]
    head, head0 = md[key], md0[key]
    tw = head.core.tower
    if big:
        assert getattr(tw, "post_units", None) is not None
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = data["train"]
    rt, at = (5e-4, 5e-5) if opt != "sgd" else (2e-5, 5e-6)
    names = CNN_PARAMS if model == "cnn_embedding" else ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if model == "cnn_embedding":
            to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, n_items, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        reg = to.extra_loss
        assert reg > 0
        got = head.train_on_batch([u, c], [y])
        got0 = head0.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        if step == 0:   # same parameters, same masks: the two losses differ by the regulariser
            assert abs((got - got0) - reg) <= 1e-5 * max(1, abs(want)), (got, got0, reg)
        np.testing.assert_allclose(head.core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in names:
            if name == "bc" or (opt != "sgd" and name in ("bd", "rm")):
                continue   # gradients that are rounding noise behind a BatchNorm
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))
    assert np.abs(tw.W.cpu().numpy() - md0[key].core.tower.W.cpu().numpy()).max() > 1e-4   # the term moved the word rows
    # loss-only steps (the monitor heads): the regulariser belongs to the loss, and the second one finds the counts clean
    e = head.core.engine
    W0 = tw.W.clone()
    for step in (3, 4):
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if model == "cnn_embedding":
            to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
        if scheme == "original":
            want = orc.step_original(u, c, y, update=False)
            got = float(head.core.step("original", e.ids(u[:b]), e.ids(c), update=False).item())
        else:
            want = (orc.step_neg_shared if scheme == "neg_shared" else orc.step_group_neg_shared)(u, c, update=False)
            got = float(head.core.step(scheme, e.ids(u), e.ids(c), update=False).item())
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
    assert torch.equal(tw.W, W0)


@pytest.mark.parametrize("model,scheme,loss,opt,act,drop", [
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", "tanh", 0.0),
    ("basic_embedding", "neg_shared", "log-loss", "sgd", "relu", 0.3),
    ("basic_embedding", "original", "skip-gram", "sgd", "sigmoid", 0.2),
    ("basic_embedding", "group_neg_shared", "log-loss", "adam", "softplus", 0.0),
    ("basic_embedding", "neg_shared", "mse", "sgd", "hard_sigmoid", 0.0),
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", "softsign", 0.2),
    ("basic_embedding", "original", "log-loss", "sgd", "linear", 0.0),
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", "softmax", 0.0),
    ("basic_embedding", "group_neg_shared", "log-loss", "sgd", "softmax", 0.2),
    ("cnn_embedding", "neg_shared", "skip-gram", "sgd", "softmax", 0.0),
    ("cnn_embedding", "neg_shared", "skip-gram", "sgd", "tanh", 0.2),
    ("cnn_embedding", "original", "skip-gram", "sgd", "linear", 0.0),
])
def test_dense_tail_variants_match_oracle(text_data, model, scheme, loss, opt, act, drop):
    """item_dense_transform['dense_hidden_actv'] other than relu and Dropout(dense_hidden_dropout) (mean_pool.py:82-97,
    cnn_model.py:114-126): three steps with word-row dropout against the float64 oracle (the kernels' masks materialised and
    fed to it) -- loss, user table, every tower parameter -- then inference-mode scores (no dropout, running statistics)."""
    if model == "cnn_embedding":
        conf = _conf_cnn(loss, scheme, wdrop=0.3, cdrop=0.2)
    else:
        conf = _conf(loss, scheme, lr=0.02 if opt != "sgd" else 0.2)
        conf.word_emb_dropout_rate = 0.3
    conf.optimizer = opt
    conf.item_dense_transform = {"dense_hidden_dim": conf.item_dim, "dense_hidden_dropout": drop, "dense_hidden_actv": act}
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), model, seed=2)
    orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, text_data, dtype=np.float64)
    orc.conf.optimizer = opt
    to.dense_act = act
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    rt, at = (5e-4, 5e-5) if opt != "sgd" else (2e-5, 5e-6)
    names = CNN_PARAMS if model == "cnn_embedding" else ("W", "Wd", "bd", "gamma", "beta", "rm", "rv")
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if model == "cnn_embedding":
            to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
        if drop:
            to.dense_mask = tw.dense_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
            assert 0.5 < (to.dense_mask > 0).mean() < 0.95 and set(np.unique(to.dense_mask)) <= {0.0, float(np.float32(1) / np.float32(1 - drop))}
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in names:
            if name == "bc" or (opt != "sgd" and name in ("bd", "rm")):
                continue   # gradients that are rounding noise behind a BatchNorm
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))
    if opt != "sgd":
        return
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float64)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("scheme,loss,opt,act,drop", [
    ("neg_shared", "skip-gram", "sgd", "relu", 0.0),          # the preset's tail minus its BatchNorm
    ("neg_shared", "log-loss", "sgd", "tanh", 0.25),
    ("original", "skip-gram", "sgd", "relu", 0.0),            # a block of > 1,024 rows? no: b (1 + k) = 256 here; row-wise launch shapes
    ("group_neg_shared", "log-loss", "sgd", "sigmoid", 0.0),
    ("neg_shared", "skip-gram", "adam", "relu", 0.0),         # Keras Adam: the Dense bias now HAS a gradient
    ("group_neg_shared", "max-margin", "sparse_adam", "softsign", 0.2),
])
def test_no_bn_tower_matches_oracle(text_data, scheme, loss, opt, act, drop):
    """conf.no_BN = True (modules/content/mean_pool.py:88-93): Dense -> Activation -> Dropout with no BatchNormalization in
    between.  Four steps with word-row dropout against the float64 oracle: loss, user table, W / Wd / bd (the Dense bias takes a
    real gradient without the BatchNorm behind it); gamma, beta and the running statistics must not move; then the
    inference-mode scores (act(Dense(h)): no statistics anywhere)."""
    conf = _conf(loss, scheme, lr=0.02 if opt != "sgd" else 0.2)
    conf.word_emb_dropout_rate = 0.3
    conf.optimizer = opt
    conf.no_BN = True
    conf.item_dense_transform = {"dense_hidden_dim": conf.item_dim, "dense_hidden_dropout": drop, "dense_hidden_actv": act}
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), "basic_embedding", seed=2)
    orc, to = _oracle(conf, md, text_data, dtype=np.float64)
    orc.conf.optimizer = opt
    to.dense_act, to.no_bn = act, True
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    assert tw.no_bn
    frozen = {n: getattr(tw, n).clone() for n in ("gamma", "beta", "rm", "rv")}
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    rt, at = (5e-4, 5e-5) if opt != "sgd" else (2e-5, 5e-6)
    bd0 = tw.bd.cpu().numpy().copy()
    for step in range(4):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if drop:
            to.dense_mask = tw.dense_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in ("W", "Wd", "bd"):
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))
    assert np.abs(tw.bd.cpu().numpy() - bd0).max() > 1e-6          # the bias really trains (it starts at 0; log-loss moves it by ~1e-5)
    assert np.abs(to.bd).max() > 1e-6
    for n, v in frozen.items():
        assert torch.equal(getattr(tw, n), v), n                     # no BatchNormalization: nothing of it moves
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float64)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=3e-4, atol=3e-5)
    # the reference's comparison is `conf.no_BN == True`: any other value keeps the BatchNormalization; the CNN tower refuses
    conf2 = _conf(loss, scheme)
    conf2.no_BN = False
    assert not getattr(get_model(conf2, get_data(dict(text_data), conf2, rand_seed=5), "basic_embedding", seed=2)["model"].core.tower, "no_bn", False)
    conf3 = _conf_cnn("skip-gram", "neg_shared")
    conf3.no_BN = True
    with pytest.raises(NotImplementedError):
        get_model(conf3, get_data(dict(text_data), conf3, rand_seed=5), "cnn_embedding", seed=2)


@pytest.mark.parametrize("model,scheme,loss,opt,content_id", [
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", False),
    ("basic_embedding", "original", "log-loss", "sgd", False),
    ("basic_embedding", "group_neg_shared", "log-loss", "adam", False),
    ("basic_embedding", "neg_shared", "skip-gram", "sgd", True),
    ("cnn_embedding", "neg_shared", "skip-gram", "sgd", False),
    ("cnn_embedding", "original", "skip-gram", "sgd", False),
])
def test_tower_without_the_dense_transform_matches_oracle(text_data, model, scheme, loss, opt, content_id):
    """conf.item_dense_transform = None (mean_pool.py:78-101 / cnn_model.py:112-130, the else branch): the pooled block -- mean of
    the word rows, or the pooled conv features -- is the item row.  Three steps with word-row dropout against the float64
    oracle (loss, user table, word table, conv weights), nothing behind the pooling moves, then inference-mode scores."""
    if model == "cnn_embedding":
        conf = _conf_cnn(loss, scheme, F=24, wdrop=0.3, cdrop=0.2)      # pooled width = num_filters = item_dim
    else:
        conf = _conf(loss, scheme, lr=0.02 if opt != "sgd" else 0.2)    # pooled width = word_dim = item_dim
        conf.word_emb_dropout_rate = 0.3
    conf.optimizer = opt
    conf.item_dense_transform = None
    conf.use_content_id = content_id
    conf.v_reg = 0.05 if content_id else 0.0
    md = get_model(conf, get_data(dict(text_data), conf, rand_seed=5), model, seed=2)
    orc, to = (_oracle_cnn if model == "cnn_embedding" else _oracle)(conf, md, text_data, dtype=np.float64)
    orc.conf.optimizer = opt
    to.no_dense = True
    head = md[{"original": "model", "neg_shared": "model_neg_shared", "group_neg_shared": "model_group_neg_shared"}[scheme]]
    tw = head.core.tower
    if content_id:
        to.Cid, to.v_reg = tw.Cid.cpu().numpy().astype(np.float64), 0.05
    dense0 = {k: getattr(tw, k).clone() for k in ("Wd", "bd", "gamma", "beta", "rm", "rv")}
    b, k = conf.batch_size_p, conf.num_negatives
    rs = np.random.RandomState(7)
    train = text_data["train"]
    rt, at = (5e-4, 5e-5) if opt != "sgd" else (2e-5, 5e-6)
    names = ("W",) + (("Wc", "g1", "b1", "rm1", "rv1") if model == "cnn_embedding" else ()) + (("Cid",) if content_id else ())
    for step in range(3):
        bt = train[rs.choice(len(train), b, replace=False)]
        if scheme == "group_neg_shared":
            bt[:, 1] = np.repeat(bt[::4, 1], 4)
        orc.row_scale = tw.dropout_scale(step).cpu().numpy().astype(np.float64)
        if model == "cnn_embedding":
            to.conv_mask = tw.conv_mask(step, b * (1 + k)).cpu().numpy().astype(np.float64)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, rs.randint(0, 260, b * k), loss)
            want = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_neg_shared(u, c)
        else:
            u, c, y = bt[:, 0], bt[:, 1], bt[:, 2]
            want = orc.step_group_neg_shared(u, c)
        got = head.train_on_batch([u, c], [y])
        assert abs(got - want) <= 1e-5 * max(1, abs(want)), (step, got, want)
        np.testing.assert_allclose(head.core.engine.U.cpu().numpy(), orc.U, rtol=rt, atol=at)
        for name in names:
            np.testing.assert_allclose(getattr(tw, name).cpu().numpy(), getattr(to, name), rtol=max(rt, 3e-4), atol=max(at, 3e-5),
                                       err_msg="%s step %d" % (name, step))
    for kname, v in dense0.items():
        assert torch.equal(getattr(tw, kname), v), kname
    if opt != "sgd":
        return
    users, items = np.arange(40), np.arange(0, 260, 3)
    got = md["model_neg_shared"].predict_on_batch([users, items])
    E = to.forward(items, train=False)
    oc2 = so.StepOracle(orc.U, E, orc.conf, dtype=np.float64)
    np.testing.assert_allclose(got, oc2.predict_neg_shared(users, np.arange(len(items))), rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("opt", ["sgd", "adam", "sparse_adam"])
@pytest.mark.parametrize("hot_words", ["1", "0"])
def test_postings_mode_equals_the_count_matrix_scan(monkeypatch, opt, hot_words):
    """train_scheme = original over a block of > 1,024 rows with word-row dropout: the postings walk (default) against the
    dense scan of the same steps -- loss bitwise (the forward is the same code), parameters to summation order; with and
    without hot words (the pad id's postings list is every item), SGD and both Adams; a SMALL-block step (dense scan) on the
    same towers in between: the count matrix must still be all-zero then, or the later steps' word rows differ."""
    monkeypatch.setenv("NNCF_HOT_WORDS", hot_words)
    data = synthetic.make_citeulike_like(users=400, items=3000, train_links=9000, seed=4, content_len=24, vocab=300)
    heads = []
    for postings in ("1", "0"):
        monkeypatch.setenv("NNCF_WORD_POSTINGS", postings)
        conf = _conf("skip-gram", "original", b=512, k=3, d=32, lr=0.01 if opt != "sgd" else 0.1)
        conf.optimizer = opt
        conf.word_emb_dropout_rate = 0.3
        md = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)
        heads.append(md)
    assert getattr(heads[0]["model"].core.tower, "post_units", None) is not None
    assert getattr(heads[1]["model"].core.tower, "post_units", None) is None
    rs = np.random.RandomState(3)
    rt, at = (5e-4, 5e-5) if opt != "sgd" else (2e-5, 5e-6)
    for step in range(4):
        bt = data["train"][rs.choice(len(data["train"]), 512, replace=False)]
        u, c, y = so.assemble_original(bt, rs.randint(0, 3000, 512 * 3), "skip-gram")
        l = [md["model"].train_on_batch([u, c], [y]) for md in heads]
        assert abs(l[0] - l[1]) <= (0 if step == 0 else 1e-6) * max(1, abs(l[1])), (step, l)   # step 0: the same forward, bitwise
        for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
            if opt != "sgd" and name in ("bd", "rm"):
                continue   # dL/dbd = sum dz is exactly 0 behind a BatchNorm: Adam turns its rounding noise into +-lr steps
            a0, a1 = (getattr(md["model"].core.tower, name).cpu().numpy() for md in heads)
            np.testing.assert_allclose(a0, a1, rtol=rt, atol=at, err_msg="%s step %d" % (name, step))
        if step == 1:   # a small block in between (neg_shared head of the same model: 512 rows, dense scan)
            l2 = [md["model_neg_shared"].train_on_batch([bt[:, 0], bt[:, 1]], [bt[:, 2]]) for md in heads]
            assert abs(l2[0] - l2[1]) <= 1e-5 * max(1, abs(l2[1])), l2


def test_the_postings_schedule_must_leave_the_hot_words_empty(monkeypatch):
    """The contract of nncf_tower_set_postings / nncf_tower_set_hot_words (include/nncf_hip.h): a hot word's gradient row comes
    from the Dense backward, so the schedule gives it an empty range -- a schedule that walks its postings as well is refused
    (it would add the row twice), and so is changing the hot set under a bound schedule."""
    from nncf_amd import _lib
    data = synthetic.make_citeulike_like(users=200, items=1500, train_links=4000, seed=4, content_len=24, vocab=300)
    conf = _conf("skip-gram", "original", b=512, k=3, d=32)
    tw = get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding", seed=2)["model"].core.tower
    assert tw.post_units is not None and tw.hot_words.size > 0
    L = _lib.lib()
    hot = np.ascontiguousarray(tw.hot_words[:1])
    assert L.nncf_tower_set_hot_words(tw.handle, hot.ctypes.data_as(_lib.c_i32p), 1) == -1
    assert b"before the postings" in L.nncf_last_error()
    # a schedule whose unit of a hot word is not empty
    units = tw.post_units.cpu().numpy().reshape(-1, 4).copy()
    first = np.flatnonzero(units[:, 0] == int(hot[0]))[0]
    assert units[first, 1] == units[first, 2]
    units[first, 2] = units[first, 1] + 1
    bad = torch.as_tensor(units).to(tw.post_units.device)
    rc = L.nncf_tower_set_postings(tw.handle, bad.data_ptr(), bad.shape[0] // 8, tw.post_item.data_ptr(), tw.slot_of.data_ptr())
    assert rc == -1 and b"empty range" in L.nncf_last_error()
    # the rejected call left the bound schedule alone: the good one is still there and binding it again is fine
    assert L.nncf_tower_set_postings(tw.handle, tw.post_units.data_ptr(), tw.post_units.shape[0] // 8, tw.post_item.data_ptr(),
                                     tw.slot_of.data_ptr()) == 0
