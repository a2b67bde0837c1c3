"""CPU: pins oracle/step_oracle.py.  The Keras/TF half of the reference cannot run here
(SURVEY.md 8(c)), so the restatement is checked against torch-CPU autograd of the loss
formulas written the way the reference writes them (utils/objectives.py), plus hand-computed
toy values.  "parity unpinned" by the reference itself -- see the oracle's header."""
import numpy as np
import pytest
import torch

from oracle import step_oracle as so

LOSSES = ["skip-gram", "mse", "log-loss", "max-margin"]


def _tables(nu=40, ni=50, d=12, seed=0):
    rs = np.random.RandomState(seed)
    return rs.uniform(-0.5, 0.5, (nu, d)), rs.uniform(-0.5, 0.5, (ni, d))


def _t_normalize(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))


def _torch_original(U, V, uid, cid, y, conf, ub=None, cb=None):
    b, k = conf.batch_size_p, conf.num_negatives
    e = U[uid]
    reg = conf.u_reg * (e * e).mean(0).sum()
    u = _t_normalize(e) if conf.normalize_user else e
    v = V[cid]
    v = _t_normalize(v) if conf.normalize_item else v
    pred = (u * v).sum(1, keepdim=True)
    if ub is not None:
        pred = pred + ub[uid].reshape(-1, 1) + cb[cid].reshape(-1, 1)     # interaction_dot.py:96-99
    y = y.reshape(-1, 1)
    lam, gamma = float(np.float32(conf.neg_loss_weight)), float(np.float32(conf.loss_gamma))
    lam_k = float(np.float32(conf.neg_loss_weight) / np.float32(k)) if k else 0.0   # objectives.py:60,66: float32 array / int
    pos, neg = pred[:b], pred[b:]
    if conf.loss == "max-margin":
        diff = pos.repeat(1, k).reshape(-1, 1) - neg
        task = torch.relu(gamma - diff).mean()
    elif conf.loss == "log-loss":
        diff = pos.repeat(1, k).reshape(-1, 1) - neg
        task = (-torch.log(torch.sigmoid(gamma * diff))).mean()
    elif conf.loss == "skip-gram":
        w = 1 + (1 - y) / 2.0 * (lam_k - 1)
        task = (-w * torch.log(torch.sigmoid(y * pred))).sum() / b
    else:
        w = 1 + (1 - y) * (lam_k - 1)
        task = (w * (y - pred) ** 2).sum() / b
    return task + reg


def _torch_neg_shared(U, V, uid, cid, conf, ub=None, cb=None):
    b = conf.batch_size_p
    e = U[uid]
    reg = conf.u_reg * (e * e).mean(0).sum()
    u = _t_normalize(e) if conf.normalize_user else e
    v = V[cid]
    v = _t_normalize(v) if conf.normalize_item else v
    pred = u @ v.t()
    if ub is not None:
        pred = pred + ub[uid].reshape(-1, 1) + cb[cid].reshape(1, -1)     # interaction_dot.py:104-107
    lam, gamma = float(np.float32(conf.neg_loss_weight)), float(np.float32(conf.loss_gamma))
    eye = torch.eye(b, dtype=pred.dtype)
    if conf.loss == "max-margin":
        diff = torch.diagonal(pred) - pred  # (b,) - (b,b): broadcasts over rows
        margin = torch.diag(torch.full((b,), -gamma, dtype=pred.dtype)) + gamma
        task = torch.relu(margin - diff).mean()
    elif conf.loss == "log-loss":
        diff = torch.diagonal(pred) - pred
        task = (-torch.log(torch.sigmoid(gamma * diff))).mean()
    elif conf.loss == "skip-gram":
        wv = float(np.float32(conf.neg_loss_weight) / np.float32(b - 1))   # objectives.py:100,107
        w = eye * (1 - wv) + wv
        label = 2 * eye - 1
        task = (-w * torch.log(torch.sigmoid(label * pred))).sum() / b
    else:
        wv = float(np.float32(conf.neg_loss_weight) / np.float32(b - 1))   # objectives.py:100,107
        w = eye * (1 - wv) + wv
        task = (w * (pred - eye) ** 2).sum() / b
    return task + reg


def _torch_group(U, V, uid, cid_u, cid_x, conf, ub=None, cb=None):
    b = conf.batch_size_p
    e = U[uid]
    reg = conf.u_reg * (e * e).mean(0).sum()
    u = _t_normalize(e) if conf.normalize_user else e
    v = V[cid_u]
    v = _t_normalize(v) if conf.normalize_item else v
    pred = u @ v.t()
    if ub is not None:
        pred = pred + ub[uid].reshape(-1, 1) + cb[cid_u].reshape(1, -1)
    nu = pred.shape[1]
    lam, gamma = float(np.float32(conf.neg_loss_weight)), float(np.float32(conf.loss_gamma))
    rows = torch.arange(b)
    pred_pos = pred[rows, cid_x].reshape(-1, 1)
    mask = torch.zeros_like(pred)
    mask[rows, cid_x] = 1
    if conf.loss == "max-margin":
        task = torch.relu(gamma - (pred_pos - pred)).mean()
    elif conf.loss == "log-loss":
        task = (-torch.log(torch.sigmoid(gamma * (pred_pos - pred)))).mean()
    elif conf.loss == "skip-gram":
        wv = lam / (nu - 1)
        label = mask * 2 - 1
        w = mask * (1 - wv) + wv
        task = (-w * torch.log(torch.sigmoid(label * pred))).sum() / b
    else:
        wv = lam / (nu - 1)
        w = mask * (1 - wv) + wv
        task = (w * (pred - mask) ** 2).sum() / b
    return task + reg


def _conf(loss, b, k, model="mf", norm=None, lr=0.05):
    gamma = {"max-margin": 0.1, "log-loss": 10.}.get(loss)
    return so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, loss_gamma=gamma, u_reg=1e-3,
                         learn_rate=lr, emb_normalization=norm, model_choice=model)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_original_vs_autograd(loss, model):
    b, k = 8, 3
    conf = _conf(loss, b, k, model)
    Un, Vn = _tables()
    rs = np.random.RandomState(1)
    pu, pi = rs.randint(0, 40, b), rs.randint(0, 50, b)
    pu[1] = pu[0]  # duplicate user across positives
    negs = rs.randint(0, 50, b * k)
    negs[0] = pi[2]  # a negative colliding with a positive item
    uid, cid, y = so.assemble_original(np.stack([pu, pi, np.ones(b, dtype=int)], 1), negs, loss)
    o = so.StepOracle(Un, Vn, conf, dtype=np.float64)
    l_o = o.step_original(uid, cid, y)
    U = torch.tensor(Un, requires_grad=True)
    V = torch.tensor(Vn, requires_grad=True)
    l_t = _torch_original(U, V, torch.tensor(uid), torch.tensor(cid), torch.tensor(y, dtype=torch.float64), conf)
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    np.testing.assert_allclose(o.last["dU"], U.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(o.last["dV"], V.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(o.U, Un - conf.learn_rate * U.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(o.V, Vn - conf.learn_rate * V.grad.numpy(), atol=1e-12)
    # fp32 mode agrees with fp64 to fp32 precision (the tolerance GPU parity is held to)
    o32 = so.StepOracle(Un, Vn, conf, dtype=np.float32)
    l32 = o32.step_original(uid, cid, y)
    assert abs(l32 - l_o) < 1e-5 * max(1, abs(l_o))
    np.testing.assert_allclose(o32.U, o.U, atol=2e-6)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_neg_shared_vs_autograd(loss, model):
    b = 8
    conf = _conf(loss, b, 0, model)
    Un, Vn = _tables(seed=2)
    rs = np.random.RandomState(3)
    uid, cid = rs.randint(0, 40, b), rs.randint(0, 50, b)
    uid[3] = uid[5]
    cid[1] = cid[6]  # duplicate item inside the batch: treated as a negative (no masking)
    o = so.StepOracle(Un, Vn, conf, dtype=np.float64)
    l_o = o.step_neg_shared(uid, cid)
    U = torch.tensor(Un, requires_grad=True)
    V = torch.tensor(Vn, requires_grad=True)
    l_t = _torch_neg_shared(U, V, torch.tensor(uid), torch.tensor(cid), conf)
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    np.testing.assert_allclose(o.last["dU"], U.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(o.last["dV"], V.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model", ["mf", "content"])
def test_group_neg_shared_vs_autograd(loss, model):
    b = 12
    conf = _conf(loss, b, 0, model)
    Un, Vn = _tables(seed=4)
    rs = np.random.RandomState(5)
    uid = rs.randint(0, 40, b)
    cid = np.repeat(rs.choice(50, 4, replace=False), 3)  # 4 groups of 3 (chop = 3)
    cid[-1] = cid[0]  # an item re-appearing later in the batch
    o = so.StepOracle(Un, Vn, conf, dtype=np.float64)
    l_o = o.step_group_neg_shared(uid, cid)
    cid_u, cid_x = so.tf_unique(cid)
    U = torch.tensor(Un, requires_grad=True)
    V = torch.tensor(Vn, requires_grad=True)
    l_t = _torch_group(U, V, torch.tensor(uid), torch.tensor(cid_u), torch.tensor(cid_x, dtype=torch.long), conf)
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    np.testing.assert_allclose(o.last["dU"], U.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(o.last["dV"], V.grad.numpy(), atol=1e-10)


def test_tf_unique_first_occurrence_order():
    u, x = so.tf_unique(np.array([7, 3, 7, 9, 3, 1]))
    assert u.tolist() == [7, 3, 9, 1] and x.tolist() == [0, 1, 0, 2, 1, 3]
    u, x = so.tf_unique(np.array([5, 5, 5]))
    assert u.tolist() == [5] and x.tolist() == [0, 0, 0]


def test_hand_computed_toys():
    # one positive + one negative, d = 1, no normalisation: skip-gram by hand
    conf = so.OracleConf(batch_size_p=1, num_negatives=1, loss="skip-gram", neg_loss_weight=4., u_reg=0., learn_rate=1.)
    o = so.StepOracle(np.array([[2.0]]), np.array([[0.5], [-1.0]]), conf, dtype=np.float64)
    loss = o.step_original([0, 0], [0, 1], [1, -1])
    s_pos, s_neg = 1.0, -2.0
    want = np.log1p(np.exp(-s_pos)) + 4.0 * np.log1p(np.exp(s_neg))  # w_neg = lambda/k = 4
    assert abs(loss - want) < 1e-12
    # neg_shared log-loss, b = 2: diff[i,j] = S[j,j] - S[i,j]; the diagonal contributes log 2
    conf = so.OracleConf(batch_size_p=2, num_negatives=0, loss="log-loss", loss_gamma=1., u_reg=0.,
                         emb_normalization=False)
    S = np.array([[1.0, 0.2], [-0.3, 0.5]])
    l, G = so.neg_shared_loss(S, conf)
    want = (2 * np.log(2) + np.log1p(np.exp(-(0.5 - 0.2))) + np.log1p(np.exp(-(1.0 + 0.3)))) / 4
    assert abs(l - want) < 1e-12
    assert abs(G.sum()) < 1e-12  # pairwise: every column's gradient sums to zero
    # group: row-anchored orientation differs from neg_shared's column-anchored one
    l2, _ = so.group_neg_shared_loss(S, np.array([0, 1]), conf)
    want2 = (2 * np.log(2) + np.log1p(np.exp(-(1.0 - 0.2))) + np.log1p(np.exp(-(0.5 + 0.3)))) / 4
    assert abs(l2 - want2) < 1e-12


def test_eval_multiple_formulas():
    truth = np.array([1, 0, 1, 0, 0, 1, 0, 0])
    pred = np.array([.9, .8, .7, .6, .5, .4, .3, .2])
    ap, rec, prec = so.eval_multiple(truth, pred, 4)
    assert abs(rec - 2 / 3) < 1e-12 and abs(prec - 0.5) < 1e-12
    assert abs(ap - (1 / 1 + 2 / 3) / 3) < 1e-12
    assert so.eval_multiple(np.zeros(8), pred, 4) == (0., 0., 0.)


def test_tower_oracle_vs_autograd_full_chain():
    """basic_embedding: MeanPool tower (mean_pool.py:46-110) + neg_shared log-loss step, every
    parameter gradient against torch autograd of the formulas as the reference wires them
    (content model on tf.unique(cid), BN batch statistics over the unique items)."""
    rs = np.random.RandomState(0)
    I, L, Wc, wd, d, nu, b = 30, 7, 20, 8, 8, 10, 8
    C = rs.randint(0, Wc, (I, L))
    C[:, :2] = 0  # left zero padding (data/readme.txt:5)
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = rs.uniform(-.1, .1, d), rs.uniform(.5, 1.5, d), rs.uniform(-.1, .1, d)
    U = rs.uniform(-.5, .5, (nu, d))
    for scheme, loss in (("neg_shared", "log-loss"), ("original", "skip-gram"), ("group_neg_shared", "max-margin")):
        k = 2 if scheme == "original" else 0
        conf = so.OracleConf(batch_size_p=b, num_negatives=k, loss=loss, u_reg=1e-3, learn_rate=0.1,
                             model_choice="basic_embedding")
        tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.1, dtype=np.float64)
        o = so.StepOracle(U, None, conf, dtype=np.float64, tower=tw)
        uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
        cid[3] = cid[0]
        if scheme == "original":
            uu, cc, yy = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=int)], 1), rs.randint(0, I, b * k), loss)
            l_o = o.step_original(uu, cc, yy)
        elif scheme == "neg_shared":
            uu, cc = uid, cid
            l_o = o.step_neg_shared(uid, cid)
        else:
            uu, cc = uid, cid
            l_o = o.step_group_neg_shared(uid, cid)
        tU = torch.tensor(U, requires_grad=True)
        tp = [torch.tensor(x, requires_grad=True) for x in (W, Wd, bd, ga, be)]
        cu, cx = so.tf_unique(cc)
        h = tp[0][torch.tensor(C[cu])].sum(1) / L
        z = h @ tp[1] + tp[2]
        mu = z.mean(0)
        var = ((z - mu) ** 2).mean(0)
        E = torch.relu(tp[3] * (z - mu) / torch.sqrt(var + 1e-3) + tp[4])

        class _V(object):  # an "item table" that is the tower output: reuse the id-table formulas
            pass
        Vt = E
        if scheme == "original":
            l_t = _torch_original(tU, Vt, torch.tensor(uu), torch.tensor(cx, dtype=torch.long),
                                  torch.tensor(yy, dtype=torch.float64), conf)
        elif scheme == "neg_shared":
            l_t = _torch_neg_shared(tU, Vt, torch.tensor(uu), torch.tensor(cx, dtype=torch.long), conf)
        else:
            cu2, cx2 = so.tf_unique(cx)
            l_t = _torch_group(tU, Vt, torch.tensor(uu), torch.tensor(cu2, dtype=torch.long),
                               torch.tensor(cx2, dtype=torch.long), conf)
        l_t.backward()
        assert abs(l_o - l_t.item()) < 1e-10, scheme
        for name, t in zip(("W", "Wd", "bd", "gamma", "beta"), tp):
            np.testing.assert_allclose(tw.grads[name], t.grad.numpy(), atol=1e-10, err_msg=scheme + name)
        np.testing.assert_allclose(o.last["dU"], tU.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("pooling,content_id,c_reg", [("max", False, 0.), ("average", True, 0.), ("max", True, 0.),
                                                      ("average", False, 0.03), ("max", True, 0.03)])
def test_tower_oracle_pooling_max_and_content_id_vs_autograd(pooling, content_id, c_reg):
    """conf.pooling = 'max' (mean_pool.py:70-72: MaxPooling1D over the whole length) and conf.use_content_id
    (mean_pool.py:102-108: tower output + Emb_Cid[cid], activity_l2(v_reg) on the Embedding output) and conf.c_reg
    (mean_pool.py:56-59: activity_l2(c_reg) on the word Embedding's output, after its row dropout): loss and every
    gradient of the oracle against torch autograd of the formulas; word-row dropout on (dropped rows are zeros that
    still take part in the maximum)."""
    rs = np.random.RandomState(1)
    I, L, Wc, wd, d, nu, b = 30, 7, 20, 8, 8, 10, 8
    C = rs.randint(0, Wc, (I, L))
    C[:, :2] = 0
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = rs.uniform(-.1, .1, d), rs.uniform(.5, 1.5, d), rs.uniform(-.1, .1, d)
    U, Cid = rs.uniform(-.5, .5, (nu, d)), rs.uniform(-.5, .5, (I, d))
    scale = ((rs.rand(Wc) < 0.7) / 0.7)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-3, learn_rate=0.1,
                         model_choice="basic_embedding")
    tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.1, dtype=np.float64)
    tw.pooling = pooling
    if content_id:
        tw.Cid, tw.v_reg = Cid.copy(), 0.05
    tw.c_reg = c_reg
    o = so.StepOracle(U, None, conf, dtype=np.float64, tower=tw)
    o.row_scale = scale
    uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
    cid[3] = cid[0]
    l_o = o.step_neg_shared(uid, cid)
    tU = torch.tensor(U, requires_grad=True)
    tp = [torch.tensor(x, requires_grad=True) for x in (W, Wd, bd, ga, be, Cid)]
    cu, cx = so.tf_unique(cid)
    rows = tp[0][torch.tensor(C[cu])] * torch.tensor(scale)[torch.tensor(C[cu])][..., None]
    h = rows.max(dim=1).values if pooling == "max" else rows.sum(1) / L
    z = h @ tp[1] + tp[2]
    mu = z.mean(0)
    var = ((z - mu) ** 2).mean(0)
    E = torch.relu(tp[3] * (z - mu) / torch.sqrt(var + 1e-3) + tp[4])
    extra = 0.
    if content_id:
        ce = tp[5][torch.tensor(cu, dtype=torch.long)]
        E = E + ce
        extra = 0.05 * (ce * ce).mean(0).sum()
    extra = extra + c_reg * (rows * rows).mean(0).sum()       # L1L2Act: sum over (L, wd) of the mean over the block's rows
    l_t = _torch_neg_shared(tU, E, torch.tensor(uid), torch.tensor(cx, dtype=torch.long), conf) + extra
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    for name, t in zip(("W", "Wd", "bd", "gamma", "beta"), tp):
        np.testing.assert_allclose(tw.grads[name], t.grad.numpy(), atol=1e-10, err_msg=name)
    np.testing.assert_allclose(o.last["dU"], tU.grad.numpy(), atol=1e-10)
    if content_id:    # the SGD step the oracle took on the id rows = lr * autograd's gradient
        np.testing.assert_allclose(Cid - tw.Cid, 0.1 * tp[5].grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "hard_sigmoid", "softplus", "softsign", "linear", "relu", "softmax"])
def test_tower_oracle_dense_tail_variants_vs_autograd(act):
    """item_dense_transform['dense_hidden_actv'] other than relu and Dropout(dense_hidden_dropout) (mean_pool.py:82-97): loss
    and every gradient of the oracle against torch autograd of Dense -> BatchNorm -> Activation -> Dropout (a given mask)."""
    rs = np.random.RandomState(2)
    I, L, Wc, wd, d, nu, b = 30, 7, 20, 8, 8, 10, 8
    C = rs.randint(0, Wc, (I, L))
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = rs.uniform(-.1, .1, d), rs.uniform(.5, 1.5, d), rs.uniform(-.1, .1, d)
    U = rs.uniform(-.5, .5, (nu, d))
    mask = (rs.rand(b, d) < 0.8) / 0.8
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="log-loss", u_reg=1e-3, learn_rate=0.1, model_choice="basic_embedding")
    tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.1, dtype=np.float64)
    tw.dense_act, tw.dense_mask = act, mask
    o = so.StepOracle(U, None, conf, dtype=np.float64, tower=tw)
    uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
    cid[3] = cid[0]
    l_o = o.step_neg_shared(uid, cid)
    tU = torch.tensor(U, requires_grad=True)
    tp = [torch.tensor(x, requires_grad=True) for x in (W, Wd, bd, ga, be)]
    cu, cx = so.tf_unique(cid)
    h = tp[0][torch.tensor(C[cu])].sum(1) / L
    z = h @ tp[1] + tp[2]
    mu = z.mean(0)
    y = tp[3] * (z - mu) / torch.sqrt(((z - mu) ** 2).mean(0) + 1e-3) + tp[4]
    f = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "hard_sigmoid": lambda v: torch.clamp(0.2 * v + 0.5, 0, 1),
         "softplus": torch.nn.functional.softplus, "softsign": torch.nn.functional.softsign, "linear": lambda v: v,
         "relu": torch.relu, "softmax": lambda v: torch.softmax(v, dim=1)}[act]
    E = f(y) * torch.tensor(mask[:len(cu)])
    l_t = _torch_neg_shared(tU, E, torch.tensor(uid), torch.tensor(cx, dtype=torch.long), conf)
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    for name, t in zip(("W", "Wd", "bd", "gamma", "beta"), tp):
        np.testing.assert_allclose(tw.grads[name], t.grad.numpy(), atol=1e-10, err_msg=name)
    np.testing.assert_allclose(o.last["dU"], tU.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("act", ["relu", "tanh", "softsign"])
def test_tower_oracle_no_bn_vs_autograd(act):
    """conf.no_BN (mean_pool.py:88-93): Dense -> Activation -> Dropout(mask) without the BatchNormalization -- loss and every
    gradient of the oracle against torch autograd; the Dense bias has a real gradient, gamma / beta / running statistics none."""
    rs = np.random.RandomState(4)
    I, L, Wc, wd, d, nu, b = 30, 7, 20, 8, 8, 10, 8
    C = rs.randint(0, Wc, (I, L))
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = rs.uniform(-.1, .1, d), rs.uniform(.5, 1.5, d), rs.uniform(-.1, .1, d)
    U = rs.uniform(-.5, .5, (nu, d))
    mask = (rs.rand(b, d) < 0.8) / 0.8
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="log-loss", u_reg=1e-3, learn_rate=0.1, model_choice="basic_embedding")
    tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.1, dtype=np.float64)
    tw.dense_act, tw.dense_mask, tw.no_bn = act, mask, True
    o = so.StepOracle(U, None, conf, dtype=np.float64, tower=tw)
    uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
    cid[3] = cid[0]
    l_o = o.step_neg_shared(uid, cid)
    tU = torch.tensor(U, requires_grad=True)
    tp = [torch.tensor(x, requires_grad=True) for x in (W, Wd, bd)]
    cu, cx = so.tf_unique(cid)
    z = (tp[0][torch.tensor(C[cu])].sum(1) / L) @ tp[1] + tp[2]
    f = {"tanh": torch.tanh, "softsign": torch.nn.functional.softsign, "relu": torch.relu}[act]
    E = f(z) * torch.tensor(mask[:len(cu)])
    l_t = _torch_neg_shared(tU, E, torch.tensor(uid), torch.tensor(cx, dtype=torch.long), conf)
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    for name, t in zip(("W", "Wd", "bd"), tp):
        np.testing.assert_allclose(tw.grads[name], t.grad.numpy(), atol=1e-10, err_msg=name)
    assert np.abs(tw.grads["bd"]).max() > 1e-4
    np.testing.assert_allclose(o.last["dU"], tU.grad.numpy(), atol=1e-10)
    for k, v in (("gamma", ga), ("beta", be), ("rm", np.zeros(d)), ("rv", np.ones(d))):   # no BatchNormalization: nothing of it moves
        assert np.array_equal(getattr(tw, k), v), k
    # inference phase: act(Dense(h)), no statistics
    Ei = tw.forward(np.arange(5), train=False)
    hi = tw.W[C[:5]].sum(1) / L
    np.testing.assert_allclose(Ei, f(torch.tensor(hi @ tw.Wd + tw.bd)).numpy(), atol=1e-12)


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_bag_form_of_the_mean_pool_oracle(opt):
    """TowerOracle.use_bag(): sum over positions == sum over distinct words x counts.  The epoch-long parity loops at the BASELINE
    shape (oracle/epoch_oracle.py) run the bag form; here it is pinned to the token-wise form (the one checked against autograd
    above) over several steps with row dropout: loss and every parameter to ~1e-13."""
    rs = np.random.RandomState(5)
    I, L, Wc, wd, d, nu, b = 120, 30, 60, 12, 12, 50, 32
    C = rs.randint(0, Wc, (I, L))
    C[:, 20:] = 0     # a pad id that fills a third of every item
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = np.zeros(d), np.ones(d), np.zeros(d)
    U = rs.uniform(-.5, .5, (nu, d))
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="skip-gram", u_reg=1e-4, learn_rate=0.05, model_choice="basic_embedding",
                         optimizer=opt)
    pair = []
    for bag in (False, True):
        tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.05, dtype=np.float64)
        if bag:
            tw.use_bag()
        pair.append((so.StepOracle(U, None, conf, dtype=np.float64, tower=tw), tw))
    for step in range(5):
        uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
        scale = (rs.rand(Wc) < 0.6) / 0.6 if step else None
        ls = []
        for o, tw in pair:
            o.row_scale = scale
            ls.append(o.step_neg_shared(uid, cid))
        assert abs(ls[0] - ls[1]) <= 1e-12 * max(1, abs(ls[0]))
    for name in ("W", "Wd", "bd", "gamma", "beta", "rm", "rv"):
        if opt == "adam" and name in ("bd", "rm"):
            # the Dense bias in front of a BatchNorm has no gradient: what Adam normalises (m / sqrt(v)) is the rounding residue
            # of sum_n dz[n] -- ~1e-17 here, a different residue for every summation order -- so bd (and the running mean that
            # carries it) drifts by ~1e-8 between two float64 evaluations of the SAME algebra; the training forward is blind to it
            assert np.abs(getattr(pair[0][1], name) - getattr(pair[1][1], name)).max() < 1e-6
            continue
        np.testing.assert_allclose(getattr(pair[0][1], name), getattr(pair[1][1], name), rtol=1e-10, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(pair[0][0].U, pair[1][0].U, rtol=1e-10, atol=1e-13)
    # (inference reads bd through z and the running mean that trails it by the momentum: Adam's residue drift shows at ~1e-8)
    tol = dict(rtol=1e-10, atol=1e-13) if opt == "sgd" else dict(rtol=0, atol=1e-6)
    np.testing.assert_allclose(pair[0][1].forward(np.arange(40), train=False), pair[1][1].forward(np.arange(40), train=False), **tol)


def test_tower_oracle_without_the_dense_transform_vs_autograd():
    """item_dense_transform falsy (mean_pool.py:78-101, the else branch): the pooled block is the item row -- loss and gradients of
    the oracle's Dense-less path (with c_reg on) against torch autograd."""
    rs = np.random.RandomState(2)
    I, L, Wc, wd, d, nu, b = 30, 7, 20, 8, 8, 10, 8
    C = rs.randint(0, Wc, (I, L))
    W, Wd = rs.uniform(-.5, .5, (Wc, wd)), rs.uniform(-.5, .5, (wd, d))
    bd, ga, be = rs.uniform(-.1, .1, d), rs.uniform(.5, 1.5, d), rs.uniform(-.1, .1, d)
    U = rs.uniform(-.5, .5, (nu, d))
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="log-loss", u_reg=1e-3, learn_rate=0.1, model_choice="basic_embedding")
    tw = so.TowerOracle(C, W, Wd, bd, ga, be, 0.1, dtype=np.float64)
    tw.no_dense, tw.c_reg = True, 0.02
    o = so.StepOracle(U, None, conf, dtype=np.float64, tower=tw)
    uid, cid = rs.randint(0, nu, b), rs.randint(0, I, b)
    cid[3] = cid[0]
    l_o = o.step_neg_shared(uid, cid)
    tU, tW = torch.tensor(U, requires_grad=True), torch.tensor(W, requires_grad=True)
    cu, cx = so.tf_unique(cid)
    rows = tW[torch.tensor(C[cu])]
    l_t = _torch_neg_shared(tU, rows.sum(1) / L, torch.tensor(uid), torch.tensor(cx, dtype=torch.long), conf) + 0.02 * (rows * rows).mean(0).sum()
    l_t.backward()
    assert abs(l_o - l_t.item()) < 1e-10
    np.testing.assert_allclose(tw.grads["W"], tW.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(o.last["dU"], tU.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(W - tw.W, 0.1 * tW.grad.numpy(), atol=1e-10)    # the SGD step it took
    for k, v in (("Wd", Wd), ("bd", bd), ("gamma", ga), ("beta", be)):        # nothing behind the pooling moves
        assert np.array_equal(getattr(tw, k), v), k


@pytest.mark.parametrize("kind", ["adam", "sparse_adam"])
def test_adam_slot_against_torch_adam(kind):
    """AdamSlot (Keras / TF formula lr_t m / (sqrt v + eps)) against torch.optim.Adam /
    SparseAdam, which differ only in where epsilon sits (eps vs eps / sqrt(1 - b2^t)): with
    eps = 1e-8 and |g| ~ 1 the two agree to ~1e-6 relative."""
    rs = np.random.RandomState(0)
    p0 = rs.randn(12, 5)
    conf = so.OracleConf(optimizer=kind, learn_rate=0.01)
    p = p0.copy()
    slot = so.AdamSlot(p)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = (torch.optim.Adam if kind == "adam" else torch.optim.SparseAdam)([tp], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 8):
        rows = np.unique(rs.randint(0, 12, 5))
        g = np.zeros_like(p)
        g[rows] = np.sign(rs.randn(len(rows), 5)) * (0.5 + rs.rand(len(rows), 5))
        slot.step(p, g, so.adam_lr_t(0.01, 0.9, 0.999, t), conf, rows if kind == "sparse_adam" else None)
        if kind == "adam":
            tp.grad = torch.tensor(g)
        else:
            tp.grad = torch.sparse_coo_tensor(torch.tensor(rows)[None], torch.tensor(g[rows]), size=p.shape)
        opt.step()
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-9)
    if kind == "sparse_adam":   # untouched rows never move, their moments never decay
        q = p0.copy()
        s2 = so.AdamSlot(q)
        g = np.zeros_like(q)
        g[3] = 1.0
        s2.step(q, g, 0.01, conf, np.array([3]))
        assert np.array_equal(np.delete(q, 3, 0), np.delete(p0, 3, 0))


def test_step_oracle_adam_moves_idle_rows_only_when_dense():
    rs = np.random.RandomState(1)
    U, V = rs.randn(20, 8).astype(np.float32), rs.randn(30, 8).astype(np.float32)
    uid, cid = rs.randint(0, 10, 16), rs.randint(0, 15, 16)
    for kind in ("adam", "sparse_adam"):
        o = so.StepOracle(U, V, so.OracleConf(batch_size_p=16, num_negatives=0, optimizer=kind), dtype=np.float32)
        o.step_neg_shared(uid, cid)
        U1 = o.U.copy()
        o.step_neg_shared(uid[::-1] * 0 + 19, cid)      # user 19 only
        idle = np.setdiff1d(np.arange(20), [19])
        moved = np.abs(o.U[idle] - U1[idle]).max()
        touched_before = np.unique(uid)
        if kind == "adam":   # momentum keeps moving the rows of step 1
            assert np.abs(o.U[touched_before] - U1[touched_before]).max() > 0
        else:
            assert moved == 0
        assert o.t == 2


@pytest.mark.parametrize("pool", ["average", "max"])
def test_cnn_tower_oracle_vs_autograd_full_chain(pool):
    """CnnTowerOracle (explicit forward / backward) against torch autograd of the same graph written
    with torch ops (conv1d 'same', batch_norm over (items, positions), relu, mask, mean | max over the positions
    (cnn_model.py:100-109), linear, batch_norm, relu), float64, through a neg_shared step."""
    rs = np.random.RandomState(3)
    n_items, L, V, wd, F, d, b, nu, fl = 40, 9, 30, 6, 7, 5, 16, 12, 3
    C = rs.randint(0, V, (n_items, L))
    W = rs.randn(V, wd) * 0.5
    Wc = rs.randn(fl, wd, F) * 0.4
    bc = rs.randn(F) * 0.1
    g1, b1 = 1 + 0.1 * rs.randn(F), 0.1 * rs.randn(F)
    Wd = rs.randn(F, d) * 0.5
    bd, gamma, beta = rs.randn(d) * 0.1, 1 + 0.1 * rs.randn(d), 0.1 * rs.randn(d)
    U = rs.randn(nu, d) * 0.5
    uid, cid = rs.randint(0, nu, b), rs.randint(0, n_items, b)
    scale = ((rs.rand(V) < 0.7) / 0.7)
    conf = so.OracleConf(batch_size_p=b, num_negatives=0, loss="log-loss", u_reg=1e-3, learn_rate=0.1,
                         model_choice="basic_embedding")
    to = so.CnnTowerOracle(C, W, Wc, bc, g1, b1, Wd, bd, gamma, beta, 0.1, dtype=np.float64)
    to.conv_pooling = pool
    cid_u, cid_x = so.tf_unique(cid)
    mask = ((rs.rand(len(cid_u), L, F) < 0.8) / 0.8)
    to.conv_mask = mask
    orc = so.StepOracle(U, None, conf, dtype=np.float64, tower=to)
    orc.row_scale = scale
    loss = orc.step_neg_shared(uid, cid)

    T = lambda x: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=True)  # noqa: E731
    tW, tWc, tbc, tg1, tb1, tWd, tbd, tga, tbe, tU = map(T, (W, Wc, bc, g1, b1, Wd, bd, gamma, beta, U))
    toks = torch.tensor(C[cid_u])
    X = tW[toks] * torch.tensor(scale)[toks][..., None]                         # (n, L, wd)
    Y = torch.nn.functional.conv1d(X.transpose(1, 2), tWc.permute(2, 1, 0), bias=tbc, padding=fl // 2).transpose(1, 2)
    mu, var = Y.mean(dim=(0, 1)), Y.var(dim=(0, 1), unbiased=False)
    a = torch.relu(tg1 * (Y - mu) / torch.sqrt(var + 1e-3) + tb1) * torch.tensor(mask)
    h = a.mean(dim=1) if pool == "average" else a.max(dim=1).values
    z = h @ tWd + tbd
    E = torch.relu(tga * (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-3) + tbe)
    lt = _torch_neg_shared(tU, E, torch.tensor(uid), torch.tensor(cid_x), conf)
    lt.backward()
    assert abs(loss - float(lt)) < 1e-9 * max(1, abs(loss))
    for name, tp in (("W", tW), ("Wc", tWc), ("bc", tbc), ("g1", tg1), ("b1", tb1), ("Wd", tWd), ("bd", tbd),
                     ("gamma", tga), ("beta", tbe)):
        np.testing.assert_allclose(to.grads[name], tp.grad.numpy(), rtol=1e-7, atol=1e-10, err_msg=name)


@pytest.mark.parametrize("loss", ["skip-gram", "mse", "log-loss", "max-margin"])
def test_interaction_bias_vs_autograd(loss):
    """interaction_bias = 'both' (interaction_dot.py:62-71,96-107): ubias[uid] + cbias[cid] added to
    every score; the oracle's bias gradients (row / column sums of dL/dscore) against torch autograd
    for the three schemes."""
    rs = np.random.RandomState(4)
    U, V = _tables()
    nu, ni, b, k = U.shape[0], V.shape[0], 8, 2
    ub0, cb0 = rs.randn(nu) * 0.1, rs.randn(ni) * 0.1
    for scheme in ("original", "neg_shared", "group_neg_shared"):
        conf = so.OracleConf(batch_size_p=b, num_negatives=k if scheme == "original" else 0, loss=loss, u_reg=1e-3,
                             learn_rate=0.1, model_choice="mf", interaction_bias="both")
        o = so.StepOracle(U, V, conf, dtype=np.float64)
        o.ub[:] = ub0
        o.cb[:] = cb0
        uid, cid = rs.randint(0, nu, b), rs.randint(0, ni, b)
        cid[3] = cid[0]
        uid[5] = uid[1]
        tU, tV, tub, tcb = [torch.tensor(x, requires_grad=True) for x in (U, V, ub0, cb0)]
        if scheme == "original":
            uu, cc, yy = so.assemble_original(np.stack([uid, cid, np.ones(b, dtype=int)], 1), rs.randint(0, ni, b * k), loss)
            lo = o.step_original(uu, cc, yy)
            lt = _torch_original(tU, tV, torch.tensor(uu), torch.tensor(cc), torch.tensor(yy, dtype=torch.float64), conf, tub, tcb)
        elif scheme == "neg_shared":
            lo = o.step_neg_shared(uid, cid)
            lt = _torch_neg_shared(tU, tV, torch.tensor(uid), torch.tensor(cid), conf, tub, tcb)
        else:
            lo = o.step_group_neg_shared(uid, cid)
            cu, cx = so.tf_unique(cid)
            lt = _torch_group(tU, tV, torch.tensor(uid), torch.tensor(cu), torch.tensor(cx), conf, tub, tcb)
        lt.backward()
        assert abs(lo - float(lt.detach())) < 1e-9 * max(1, abs(lo)), (scheme, lo, float(lt.detach()))
        np.testing.assert_allclose(o.ub, ub0 - 0.1 * tub.grad.numpy(), rtol=1e-9, atol=1e-12, err_msg=scheme)
        np.testing.assert_allclose(o.cb, cb0 - 0.1 * tcb.grad.numpy(), rtol=1e-9, atol=1e-12, err_msg=scheme)
        np.testing.assert_allclose(o.U, U - 0.1 * tU.grad.numpy(), rtol=1e-8, atol=1e-11, err_msg=scheme)


def test_sparse_adam_as_written_by_hand():
    """utils/optimizer.py:108-134 with a duplicated index, step 1 from zero moments: index 2 occurs twice (gradients
    g_a then g_b): m = (1-b1) g_b, v = (1-b2) g_b^2, and the step lr_t m / (sqrt(v) + eps) is subtracted TWICE;
    index 0 occurs once.  The summed form would use g_a + g_b and subtract once."""
    conf = so.OracleConf(optimizer="sparse_adam", learn_rate=0.1)
    p = np.zeros((3, 2), dtype=np.float64)
    slot = so.AdamSlot(p)
    g = np.array([[1.0, -2.0], [0.5, 4.0], [-3.0, 0.25]])
    idx = np.array([2, 0, 2])
    lr_t = so.adam_lr_t(0.1, conf.beta1, conf.beta2, 1)
    slot.step_as_written(p, idx, g, lr_t, conf)
    b1, b2, eps = conf.beta1, conf.beta2, conf.epsilon
    for row, glast, cnt in ((2, g[2], 2), (0, g[1], 1)):
        m, v = (1 - b1) * glast, (1 - b2) * glast ** 2
        np.testing.assert_allclose(slot.m[row], m, rtol=1e-12)
        np.testing.assert_allclose(slot.v[row], v, rtol=1e-12)
        np.testing.assert_allclose(p[row], -cnt * lr_t * m / (np.sqrt(v) + eps), rtol=1e-12)
    assert np.all(p[1] == 0) and np.all(slot.m[1] == 0)
