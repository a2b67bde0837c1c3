"""CPU: the filter-bank CNN tower oracle (oracle/step_oracle.py CnnBankTowerOracle, modules/content/cnn_model.py:63-81)
against the single-convolution oracle it generalises: with one bank they are the same model; a bank [1, 3, 5] is ONE
'same' convolution of length 5 whose shorter filters are centred and zero-padded -- forward, every gradient on the
structural taps, and the parameters after an SGD step agree."""
import numpy as np

from oracle import step_oracle as so


def _params(rs, fls, wd, F):
    return ([rs.randn(fl, wd, F) * 0.3 for fl in fls], [rs.randn(F) * 0.1 for _ in fls],
            [1 + rs.randn(F) * 0.1 for _ in fls], [rs.randn(F) * 0.1 for _ in fls])


def test_one_bank_is_the_single_convolution_oracle():
    rs = np.random.RandomState(0)
    nI, L, nw, wd, F, d = 12, 9, 30, 6, 4, 5
    C, W = rs.randint(0, nw, (nI, L)), rs.randn(nw, wd) * 0.3
    Wc, bc, g1, b1 = _params(rs, [3], wd, F)
    Wd, bd, gam, bet = rs.randn(F, d) * 0.3, rs.randn(d) * 0.1, 1 + rs.randn(d) * 0.1, rs.randn(d) * 0.1
    a = so.CnnTowerOracle(C, W, Wc[0], bc[0], g1[0], b1[0], Wd, bd, gam, bet, 0.1, dtype=np.float64)
    b = so.CnnBankTowerOracle(C, W, Wc, bc, g1, b1, Wd, bd, gam, bet, 0.1, dtype=np.float64)
    ids = np.array([3, 1, 7, 2])
    mask = (rs.rand(4, L, F) < 0.7) / 0.7
    a.conv_mask, b.conv_mask = mask, [mask]
    np.testing.assert_allclose(a.forward(ids), b.forward(ids), rtol=1e-12)
    dA = rs.randn(4, d)
    a.backward_update(dA)
    b.backward_update(dA)
    for k in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
        np.testing.assert_allclose(getattr(a, k), getattr(b, k), rtol=1e-12, err_msg=k)
    for k in ("Wc", "bc", "g1", "b1", "rm1", "rv1"):
        np.testing.assert_allclose(getattr(a, k), getattr(b, k)[0], rtol=1e-12, err_msg=k)
    np.testing.assert_allclose(a.forward(ids, train=False), b.forward(ids, train=False), rtol=1e-12)


def test_bank_is_a_zero_padded_convolution_of_the_longest_length():
    rs = np.random.RandomState(1)
    nI, L, nw, wd, F, d = 12, 9, 30, 6, 4, 5
    C, W = rs.randint(0, nw, (nI, L)), rs.randn(nw, wd) * 0.3
    fls = [1, 3, 5]
    Wc, bc, g1, b1 = _params(rs, fls, wd, F)
    Wd, bd, gam, bet = rs.randn(3 * F, d) * 0.3, rs.randn(d) * 0.1, 1 + rs.randn(d) * 0.1, rs.randn(d) * 0.1
    b = so.CnnBankTowerOracle(C, W, Wc, bc, g1, b1, Wd, bd, gam, bet, 0.1, dtype=np.float64)
    big = np.zeros((5, wd, 3 * F))
    for g, fl in enumerate(fls):
        p = (5 - fl) // 2
        big[p:p + fl, :, g * F:(g + 1) * F] = Wc[g]
    a = so.CnnTowerOracle(C, W, big, np.concatenate(bc), np.concatenate(g1), np.concatenate(b1), Wd, bd, gam, bet, 0.1,
                          dtype=np.float64)
    ids = np.array([3, 1, 7, 2, 11])
    masks = [(rs.rand(5, L, F) < 0.7) / 0.7 for _ in fls]
    b.conv_mask, a.conv_mask = masks, np.concatenate(masks, axis=2)
    scale = (rs.rand(nw) < 0.8) / 0.8
    np.testing.assert_allclose(a.forward(ids, row_scale=scale), b.forward(ids, row_scale=scale), rtol=1e-12)
    dA = rs.randn(5, d)
    a.backward_update(dA)
    b.backward_update(dA)
    for k in ("W", "Wd", "bd", "gamma", "beta"):
        np.testing.assert_allclose(getattr(a, k), getattr(b, k), rtol=1e-10, err_msg=k)
    for g, fl in enumerate(fls):
        p = (5 - fl) // 2
        np.testing.assert_allclose(a.Wc[p:p + fl, :, g * F:(g + 1) * F], b.Wc[g], rtol=1e-10)
        for k in ("bc", "g1", "b1", "rm1", "rv1"):
            np.testing.assert_allclose(getattr(a, k)[g * F:(g + 1) * F], getattr(b, k)[g], rtol=1e-10, err_msg=k)
