"""CPU: host logic of the path (conf surface, degree histogram, stratified shuffle, truth CSR,
synthetic data) and the oracle's metric code against vectors produced by the reference's own
Python (tools/gen_golden_host.py)."""
import os

import numpy as np
import pytest

from oracle import step_oracle as so
from nncf_amd import synthetic
from nncf_amd.conf import get_conf
from nncf_amd.data_utils import degree_dist, group_shuffle_train

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_eval_multiple_matches_reference_outputs(gold):
    for c in gold["ev_cases"].tolist():
        got = so.eval_multiple(gold["ev%d_truth" % c], gold["ev%d_pred" % c], int(gold["ev%d_topk" % c][0]))
        np.testing.assert_allclose(np.array(got), gold["ev%d_out" % c], rtol=0, atol=1e-12)


def test_group_shuffle_matches_reference_up_to_within_group_order(gold):
    for c in gold["gs_cases"].tolist():
        train, want = gold["gs%d_train" % c], gold["gs%d_out" % c]
        col = int(gold["gs%d_by" % c][0])
        iidx = np.arange(train[:, col].max() + 1)
        np.random.seed(int(gold["gs%d_seed" % c][0]))
        got = group_shuffle_train(train.copy(), by="user" if col == 0 else "item", chop=0, iidx=iidx)
        assert got.shape == want.shape
        # identical sequence of groups and identical group membership (the reference's unstable
        # argsort may order rows inside a group differently)
        assert np.array_equal(got[:, col], want[:, col])
        for g in np.unique(got[:, col]):
            a, b = got[got[:, col] == g], want[want[:, col] == g]
            assert np.array_equal(a[np.lexsort(a.T)], b[np.lexsort(b.T)])


def test_group_shuffle_equals_reference_with_stable_argsort_exactly(gold):
    """chop in {0, 2, 5}, by item and by user, two consecutive epochs (iidx persists shuffled): the build's
    group_shuffle_train == the reference's function text run here with `.argsort()` -> `.argsort(kind='stable')`
    (tools/gen_golden_host.py), row for row: same NumPy stream, same permutations."""
    for c in gold["gst_cases"].tolist():
        train = gold["gst%d_train" % c]
        seed, col, chop = [int(x) for x in gold["gst%d_cfg" % c]]
        iidx = np.arange(train[:, col].max() + 1)
        np.random.seed(seed)
        got = group_shuffle_train(train.copy(), by="user" if col == 0 else "item", chop=chop, iidx=iidx)
        assert np.array_equal(got, gold["gst%d_out" % c]), (c, chop)
        got2 = group_shuffle_train(got.copy(), by="user" if col == 0 else "item", chop=chop, iidx=iidx)
        assert np.array_equal(got2, gold["gst%d_out2" % c]), (c, chop)


def test_group_shuffle_chop_properties():
    rs = np.random.RandomState(0)
    n, chop = 1003, 5
    train = np.stack([rs.randint(0, 90, n), rs.randint(0, 40, n), np.ones(n, dtype=np.int64)], 1)
    iidx = np.arange(40)
    rng = np.random.RandomState(3)
    out = group_shuffle_train(train.copy(), by="item", chop=chop, iidx=iidx, rng=rng)
    assert sorted(map(tuple, out)) == sorted(map(tuple, train))  # a permutation of the links
    # iidx is shuffled in place and persists across epochs (data_utils.py:231)
    assert not np.array_equal(iidx, np.arange(40)) and sorted(iidx) == list(range(40))
    # every aligned block of `chop` rows spans at most 2 items... no: a block is a contiguous run
    # of the group-sorted array, so its items are consecutive in group order
    key = iidx[out[:, 1]]
    bulk = (n // chop) * chop
    for s in range(0, bulk, chop):
        blk = key[s:s + chop]
        assert np.all(np.diff(blk) >= 0)
    assert np.all(np.diff(key[bulk:]) >= 0)  # remainder rows stay at the end, in group order


def test_degree_dist_follows_get_sampler():
    train = np.array([[0, 3, 1], [1, 3, 1], [2, 0, 1], [2, 5, 1]])
    assert degree_dist(train, 1, "unigram").tolist() == [1, 0, 0, 2, 0, 1]
    assert degree_dist(train, 1, "uniform_no_correction").tolist() == [1, 0, 0, 1, 0, 1]  # suffix ignored (:201)
    assert degree_dist(train, 0, "unigram").tolist() == [1, 1, 2]
    with pytest.raises(AssertionError):
        degree_dist(train, 1, "zipf")


def test_conf_defaults_and_override_semantics():
    c = get_conf("x", "default", None, model_choice="basic_embedding")
    assert (c.batch_size_p, c.num_negatives, c.loss, c.learn_rate, c.loss_gamma, c.neg_loss_weight) == \
        (512, 10, "skip-gram", 0.01, 10, 128)
    assert c.emb_normalization is False and c.chop_size == 2 and c.eval_topk == 50 and c.u_reg == 1e-6
    # learn_rate / gamma / weight are NOT recomputed from an overridden loss (basic_embedding_conf.py:22-24)
    c = get_conf("x", "default", {"loss": "max-margin"}, model_choice="basic_embedding")
    assert c.loss_gamma == 10 and c.emb_normalization is True
    # 'best' forces num_negatives = 10 unless reset_after_getconf re-applies param_dict (:137-141)
    c = get_conf("x", "best", {"num_negatives": 0}, model_choice="mf")
    assert c.num_negatives == 10 and c.max_epoch == 30
    c = get_conf("x", "best", {"num_negatives": 0, "reset_after_getconf": True}, model_choice="mf")
    assert c.num_negatives == 0
    assert get_conf("x", "default", None, model_choice="mf").interaction_bias == "item"
    with pytest.raises(AssertionError):
        get_conf("x", "nope")


def test_truth_csr_equals_dense_matrix():
    from nncf_amd.objectives import _truth_csr
    rs = np.random.RandomState(2)
    links = np.stack([rs.randint(0, 30, 200), rs.randint(0, 50, 200), np.ones(200, dtype=np.int64)], 1)
    cand = np.unique(links[:150, 1])
    ptr, idx = _truth_csr(links, cand, 30)
    dense = np.zeros((30, len(cand)), dtype=np.int32)
    slot = {it: i for i, it in enumerate(cand)}
    for u, i, _ in links:  # objectives.py:412-416
        if i in slot:
            dense[u, slot[i]] = 1
    for u in range(30):
        assert np.array_equal(np.nonzero(dense[u])[0], idx[ptr[u]:ptr[u + 1]])


def test_synthetic_schema():
    d = synthetic.make_citeulike_like(users=300, items=200, train_links=3000, seed=1, content_len=12, vocab=50)
    assert set(d) == {"C", "test_items", "train_items", "test", "test_seen", "train"}  # data/readme.txt:3
    assert d["train"].shape[1] == 3 and d["train"].dtype == np.int64 and np.all(d["train"][:, 2] == 1)
    assert len(np.unique(d["train"][:, :2], axis=0)) == len(d["train"])      # distinct links
    assert not set(d["train_items"]) & set(d["test_items"])                    # cold-item split
    assert set(np.unique(d["train"][:, 1])) == set(d["train_items"])          # every train item has a link
    assert set(np.unique(d["test"][:, 1])) <= set(d["test_items"])
    assert len(np.unique(d["train"][:, 0])) == 300                            # every user trains
    C = d["C"]
    assert C.shape == (200, 12) and C.min() == 0
    first_nonzero = (C != 0).argmax(axis=1)
    for r in range(200):  # zero padding at the beginning (data/readme.txt:5)
        assert np.all(C[r, first_nonzero[r]:] != 0)
    spec = synthetic.data_spec_of(d)
    assert (spec.user_count, spec.item_count, spec.max_content_len) == (300, 200, 12)


def test_group_sampler_matches_reference_golden():
    """nncf_amd.data_utils.GroupSampler against batches the reference's own class produced here
    (tools/gen_golden_group_sampler.py): same sampler streams (CPU checker of the B1 sampler
    injected), same NumPy / random seeds -> identical batches."""
    import random
    from oracle import sampler_oracle as so
    from nncf_amd.data_utils import GroupSampler, degree_dist
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "group_sampler_golden.npz"))
    train = g["train"]

    def factory(ratings, neg_dist, neg_sampling_power, column, batch_mode):
        s = so.OracleSampler(degree_dist(ratings, column, neg_dist), neg_sampling_power, 0)
        return s.sample_batch

    names = ["unigram", "uniform", "uniform_no_correction"]
    for case in range(int(g["n_cases"][0])):
        chop, nd = [int(x) for x in g["c%d_cfg" % case]]
        gs = GroupSampler(train, group_by="item", chop=chop, neg_dist=names[nd], neg_sign=-1, sampler_factory=factory)
        np.random.seed(100 + case)
        random.seed(200 + case)
        assert np.array_equal(gs.sample(64), g["c%d_a" % case]), case
        assert np.array_equal(gs.sample(50), g["c%d_b" % case]), case
        assert np.array_equal(gs.sample_with_negs(32, 3), g["c%d_c" % case]), case
        assert np.array_equal(gs.sample_with_negs(40, 2), g["c%d_d" % case]), case
    gs = GroupSampler(train, group_by="user", chop=3, neg_dist="unigram", neg_sign=0, sampler_factory=factory)
    np.random.seed(999)
    random.seed(998)
    assert np.array_equal(gs.sample(30), g["u_a"])
    assert np.array_equal(gs.sample_with_negs(30, 2), g["u_c"])


def test_shuffle_rows_is_np_random_shuffle():
    """train_base.shuffle_rows: the permutation np.random.shuffle would apply to the rows, the same RNG
    consumption, in place."""
    from nncf_amd.train_base import shuffle_rows
    a = np.arange(3000).reshape(1000, 3).astype(np.int64)
    b = a.copy()
    np.random.seed(11)
    np.random.shuffle(a)
    after_a = np.random.randint(1 << 30)
    np.random.seed(11)
    shuffle_rows(b)
    after_b = np.random.randint(1 << 30)
    assert np.array_equal(a, b) and after_a == after_b


def test_cnn_front_spec_from_the_reference_presets():
    """model_framework._conv_spec: what of cnn_embedding_conf.py's conv settings the build runs -- the `default`
    preset's filter bank [[1, 3, 5]] x 50 (cnn_embedding_conf.py:42-45) and the `best` preset's single length 3
    (:129), average or max pooling over the whole length (:100-109); stacked layers, even lengths, pooling windows shorter than
    the text and gating are refused with NotImplementedError."""
    import pytest
    from nncf_amd.conf import get_conf
    from nncf_amd.model_framework import _conv_spec
    d = get_conf("synthetic_citeulike", "default", None, model_choice="cnn_embedding")
    assert (d.filter_lengths, d.num_filters, d.word_dim, d.item_dim) == ([[1, 3, 5]], [50], 50, 50)
    assert _conv_spec(d) == dict(filters=50, filter_len=1, filter_lens=[1, 3, 5], dropout=float(d.conv_dropout_rate), pooling="average")
    b = get_conf("synthetic_citeulike", "best", None, model_choice="cnn_embedding")
    spec = _conv_spec(b)
    assert spec["filter_lens"] == [int(b.filter_lengths[0])] and spec["filters"] == int(b.num_filters[0])
    for bad in ({"filter_lengths": [3, 3], "num_filters": [50, 50], "poolings": ["average", "average"], "pool_lengths": [-1, -1]},
                {"filter_lengths": [[2, 3]]}, {"filter_lengths": [[1, 3, 5, 7, 9]]}, {"poolings": ["min"]}, {"pool_lengths": [4]},
                {"contextual_spatial_gated_input": {"gating_hidden_dim": 3}}):
        # 'best' + reset_after_getconf: the override lands after _post_init (which resets the gating fields, as the
        # reference's does: cnn_embedding_conf.py:87-90)
        c = get_conf("synthetic_citeulike", "best", dict({"reset_after_getconf": True}, **bad), model_choice="cnn_embedding")
        with pytest.raises(NotImplementedError):
            _conv_spec(c)
    mx = get_conf("synthetic_citeulike", "best", {"reset_after_getconf": True, "poolings": ["max"]}, model_choice="cnn_embedding")
    assert _conv_spec(mx)["pooling"] == "max"


def test_postings_schedule_invariants():
    """nncf_amd.mean_pool.postings_schedule (the static schedule behind nncf_tower_set_postings): every word exactly once as
    a first unit, a word's units contiguous in ONE workgroup and covering its postings in order, hot words with an empty
    range, units of at most 512 postings unless the word needs more than 8 of them, at most 8 units per workgroup."""
    from nncf_amd.mean_pool import postings_schedule
    rs = np.random.RandomState(0)
    V = 700
    df = np.minimum((3000 * rs.power(0.2, V)).astype(np.int64), 3000)
    df[0], df[5], df[9] = 9000, 4100, 0            # needs > 8 units / exactly 8 + 1 / no postings at all
    pptr = np.concatenate([[0], np.cumsum(df)])
    for hot in ((), (0, 3)):
        units, n_wg = postings_schedule(pptr, hot)
        assert units.dtype == np.int32 and units.shape == (n_wg * 8, 4)
        firsts = units[units[:, 3] > 0]
        assert sorted(firsts[:, 0].tolist()) == list(range(V))
        for g in range(n_wg):
            blk = units[g * 8:(g + 1) * 8]
            j = 0
            while j < 8:
                w, b, e, k = (int(v) for v in blk[j])
                if w < 0:
                    assert (b, e, k) == (0, 0, 0)
                    j += 1
                    continue
                assert k > 0 and j + k <= 8 and b == pptr[w]
                end = pptr[w] if w in hot else pptr[w + 1]
                assert int(blk[j + k - 1][2]) == end
                for i in range(1, k):
                    assert int(blk[j + i][0]) == w and int(blk[j + i][3]) == 0 and int(blk[j + i][1]) == int(blk[j + i - 1][2])
                lens = blk[j:j + k, 2] - blk[j:j + k, 1]
                assert lens.max() <= 512 or k == 8
                j += k


def test_bench_prices_the_large_batch_kernels_on_the_f16_pipe():
    """bench.py price_on_f16_pipe: the large-batch kernels' products run on the f16 matrix pipe -- `peak` is that pipe's dense peak,
    `frac` = achieved / peak stays a fraction (the fp32-equivalent flop against the fp32-MFMA peak reaches 1 since round 5 and is
    kept beside it), and the issued f16 flop (4 x algorithmic) is named."""
    import bench
    r = {"kernel": "fused_wide_kernel", "bound": "mfma", "achieved": 154.0, "peak": bench.PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": 154.0 / bench.PEAK_F32_MFMA_TFLOPS, "frac_uncorrected_bracket": 0.97, "algorithmic_per_launch": 51.54e9, "kernel_ms": 0.3354}
    bench.price_on_f16_pipe(r, True)
    assert r["peak"] == bench.PEAK_F16_MFMA_TFLOPS and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["frac_of_fp32_mfma_peak"] - 154.0 / bench.PEAK_F32_MFMA_TFLOPS) < 1e-12
    assert abs(r["issued_f16_tflops"] - 4 * 51.54e9 / 0.3354e-3 / 1e12) < 1e-6
    assert 0 < r["frac_issued_of_f16_mfma_peak"] < 1 and r["pipe"].startswith("f16 MFMA")
    r2 = {"kernel": "fused_wide_kernel", "achieved": 77.0, "peak": bench.PEAK_F32_MFMA_TFLOPS, "frac": 0.49, "algorithmic_per_launch": 51.54e9,
          "kernel_ms": 0.666}
    bench.price_on_f16_pipe(r2, False)      # NNCF_WIDE_F16SPLIT=0: the fp32 pipe, priced as before
    assert r2["peak"] == bench.PEAK_F32_MFMA_TFLOPS and r2["frac"] == 0.49 and r2["pipe"].startswith("f32 MFMA")


def test_dense_less_tower_needs_the_pooled_width_to_be_item_dim():
    """conf.item_dense_transform = None (mean_pool.py:78-101, the else branch): the pooled block is the item row, so word_dim (or
    the conv filters) must equal item_dim -- checked in get_model before any device work."""
    from nncf_amd.data_utils import get_data
    from nncf_amd.model_framework import get_model
    data = synthetic.make_citeulike_like(users=50, items=40, train_links=400, seed=3, content_len=12, vocab=30)
    conf = get_conf("synthetic", "best", {"reset_after_getconf": True, "word_dim": 16, "item_dim": 24, "user_dim": 24}, model_choice="basic_embedding")
    conf.item_dense_transform = None
    with pytest.raises(ValueError, match="pooled width"):
        get_model(conf, get_data(dict(data), conf, rand_seed=5), "basic_embedding")
    # conf.no_BN is built for the mean-pool tower since round 6 (tests/test_tower_gpu.py::test_no_bn_tower_matches_oracle);
    # the CNN tower, whose convolution BatchNorm it would drop too (cnn_model.py:93), still refuses -- before any device work
    conf2 = get_conf("synthetic", "best", {"reset_after_getconf": True}, model_choice="cnn_embedding")
    conf2.no_BN = True
    with pytest.raises(NotImplementedError):
        get_model(conf2, get_data(dict(data), conf2, rand_seed=5), "cnn_embedding")
