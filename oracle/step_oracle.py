"""TEST INFRASTRUCTURE ONLY -- numpy restatement (oracle) of the reference's per-batch
scoring path: embedding gather -> (L2 normalise) -> dot / U.V^T interaction -> loss ->
analytic gradient -> sparse SGD scatter update.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; nncf_amd never does.

Follows (all paths under /root/reference):
  models/model_framework.py:36-176   graph wiring: gathers, tf.unique, slicing, heads
  utils/objectives.py:15-220         the four loss families x {skip-gram, mse, log-loss, max-margin}
  modules/interaction/interaction_dot.py:82-109   'mul' (row-wise dot) and 'matmul' (U.V^T)
  utils/utilities.py:74-83,122-135   K.diag via gather_nd; L1L2Act activity regulariser
  models/train_{original,neg_shared,group_neg_shared}.py   batch assembly conventions

Parity status.  PINNED to the reference's own text: the loss factories, InteractionDot.call and the
activity regulariser (objectives.py:15-220, interaction_dot.py:82-107, utilities.py:122-135) are
exec'd from /root/reference by tools/gen_golden_losses.py over a primitive-op K / tf shim; their
losses and autograd gradients (tests/golden/loss_golden.npz: 4 families x 4 losses x 7 shapes,
float32 and float64) are what tests/test_loss_golden.py holds this file to (1e-9 in float64).
NOT pinnable here (Keras 1.2.2 / TF 1.0 can neither be imported nor installed, SURVEY.md 8(c);
the reference ships no tests or vectors) -- items taken from memory of the absent packages,
marked [K-mem]:
  * tf.nn.l2_normalize(x, dim=-1) = x * rsqrt(max(sum(x^2), 1e-12))            [K-mem]
  * a custom-objective Keras model's train_on_batch returns mean(loss_tensor) + sum of
    activity-regulariser terms of the layers on the model's own graph             [K-mem]
  * K.relu gradient at 0 is 0; K.repeat's (samples, n, dim) layout; tf.unique order [K-mem]
  * BatchNormalization eps 1e-3 / momentum 0.99, Embedding / Dense initialisers, Embedding row
    dropout, the Keras Adam formula (text towers and optimisers below)            [K-mem]
The optimiser is the north-star's sparse SGD (BASELINE.json), not the reference's dense
Adam (configs/basic_embedding_conf.py:59-60): table[row] -= lr * sum_of_row_gradients.
-log(sigmoid(x)) is evaluated as softplus(-x) (identical wherever the reference's
K.log(K.sigmoid(x)) is finite).
"""
import numpy as np

LOSSES = ("skip-gram", "mse", "log-loss", "max-margin")


class OracleConf(object):
    """Subset of configs/basic_embedding_conf.py:10-56 consumed on the path."""

    def __init__(self, batch_size_p=512, num_negatives=10, loss="skip-gram", loss_gamma=None,
                 neg_loss_weight=None, u_reg=1e-6, learn_rate=0.01, emb_normalization=None,
                 model_choice="mf", optimizer="sgd", beta1=0.9, beta2=0.999, epsilon=1e-8, interaction_bias=None):
        self.batch_size_p = int(batch_size_p)
        self.num_negatives = int(num_negatives)
        assert loss in LOSSES, "[ERROR!] loss %s not specified." % loss
        self.loss = loss
        # basic_embedding_conf.py:22-24 defaults
        self.loss_gamma = (0.1 if loss == "max-margin" else 10.) if loss_gamma is None else float(loss_gamma)
        self.neg_loss_weight = (8. if loss == "mse" else 128.) if neg_loss_weight is None else float(neg_loss_weight)
        self.u_reg = float(u_reg)
        self.learn_rate = float(learn_rate)
        if emb_normalization is None:  # basic_embedding_conf.py:52-56
            emb_normalization = loss in ("max-margin", "log-loss")
        self.emb_normalization = bool(emb_normalization)
        self.model_choice = model_choice
        # 'sgd' (the build's sparse SGD) | 'adam' (Keras Adam, basic_embedding_conf.py:59-60: dense)
        # | 'sparse_adam' (utils/optimizer.py:108-134; duplicate ids summed first)
        assert optimizer in ("sgd", "adam", "sparse_adam")
        self.optimizer, self.beta1, self.beta2, self.epsilon = optimizer, beta1, beta2, epsilon
        self.sparse_adam_duplicates = "sum"     # | 'as_written' (AdamSlot.step_as_written)
        # InteractionDot bias (interaction_dot.py:48-49,62-71): None | 'user' | 'item' | 'both'; zero init
        assert interaction_bias in ("user", "item", "both", None), "ERROR! Unknown interation bias"
        self.interaction_bias = interaction_bias
        # model_framework.py:62-63 normalises U for every model; :109-111 normalises the item
        # side only inside the content-model branch (not for 'mf', :85-88).
        self.normalize_user = self.emb_normalization
        self.normalize_item = self.emb_normalization and model_choice != "mf"


# ----------------------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------------------
def _softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def _f32_const(v, dt):
    """gamma / lambda as the reference holds them: np.array(v, dtype=np.float32) (objectives.py:23,32).
    Identity for the float32 oracle; keeps the float64 oracle on the reference's constants."""
    return dt.type(np.float32(v))


def _f32_ratio(v, n, dt):
    """np.float32 array / python int stays float32 (objectives.py:60,66,100,107,143,151)."""
    return dt.type(np.float32(v) / np.float32(n))


def _sigmoid(x):
    # numerically stable logistic, same dtype as x
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(x.dtype)


def l2_normalize(x):
    """tf.nn.l2_normalize(x, dim=-1) (model_framework.py:63,110-111) [K-mem]."""
    ss = np.sum(x * x, axis=-1, keepdims=True)
    inv = (1.0 / np.sqrt(np.maximum(ss, x.dtype.type(1e-12)))).astype(x.dtype)
    return x * inv, inv


def l2_normalize_backward(xhat, inv, g):
    """d/dx of xhat = x*inv:  (g - xhat (xhat.g)) * inv   (valid where sum(x^2) > 1e-12)."""
    return (g - xhat * np.sum(xhat * g, axis=-1, keepdims=True)) * inv


def tf_unique(x):
    """tf.unique (model_framework.py:45-48): unique values in FIRST-OCCURRENCE order and the
    index of each element into them."""
    x = np.asarray(x)
    _, first, inv = np.unique(x, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # sorted-unique slot -> first-occurrence rank
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return x[np.sort(first)], rank[inv].astype(np.int32)


# ----------------------------------------------------------------------------------------
# losses: value and d(loss)/d(score)
# ----------------------------------------------------------------------------------------
def original_loss(s, y, conf):
    """get_original_loss (objectives.py:35-75).  s, y: (b(1+k),) ; first b rows positive,
    negative row j (0-based among negatives) belongs to positive j // k (K.repeat, :51-57)."""
    b, k = conf.batch_size_p, conf.num_negatives
    dt = s.dtype
    lam, gamma = _f32_const(conf.neg_loss_weight, dt), _f32_const(conf.loss_gamma, dt)
    g = np.zeros_like(s)
    if conf.loss in ("max-margin", "log-loss"):
        pos, neg = s[:b], s[b:]
        diff = np.repeat(pos, k) - neg
        m = dt.type(b * k)
        if conf.loss == "max-margin":
            loss = np.sum(np.maximum(gamma - diff, 0)) / m           # K.mean(K.relu(margin - diff))
            gd = np.where(gamma - diff > 0, dt.type(-1), dt.type(0)) / m
        else:
            loss = np.sum(_softplus(-gamma * diff)) / m              # K.mean(-log sigmoid(g*diff))
            gd = -gamma * _sigmoid(-gamma * diff) / m
        g[b:] = -gd
        g[:b] = gd.reshape(b, k).sum(axis=1)
    elif conf.loss == "skip-gram":
        wv = _f32_ratio(conf.neg_loss_weight, k, dt)
        w = 1 + (1 - y) / 2 * (wv - 1)                               # objectives.py:60-62
        loss = np.sum(w * _softplus(-y * s)) / dt.type(b)
        g = -w * y * _sigmoid(-y * s) / dt.type(b)
    else:  # mse
        wv = _f32_ratio(conf.neg_loss_weight, k, dt)
        w = 1 + (1 - y) * (wv - 1)                                   # objectives.py:66-68
        loss = np.sum(w * (y - s) ** 2) / dt.type(b)
        g = 2 * w * (s - y) / dt.type(b)
    return dt.type(loss), g.astype(dt)


def neg_shared_loss(S, conf):
    """get_neg_shared_loss (objectives.py:78-117).  S: (b,b), diagonal = positives.
    Pairwise orientation follows the CODE: K.diag(y_pred) - y_pred is (b,) - (b,b), i.e.
    diff[i,j] = S[j,j] - S[i,j]  (objectives.py:92,97; utilities.py:74-83)."""
    b = S.shape[0]
    dt = S.dtype
    lam, gamma = _f32_const(conf.neg_loss_weight, dt), _f32_const(conf.loss_gamma, dt)
    eye = np.eye(b, dtype=bool)
    if conf.loss in ("max-margin", "log-loss"):
        diff = np.diag(S)[None, :] - S
        m = dt.type(b * b)
        if conf.loss == "max-margin":
            margin = np.where(eye, dt.type(0), gamma)                 # objectives.py:93
            act = margin - diff
            loss = np.sum(np.maximum(act, 0)) / m
            gd = np.where(act > 0, dt.type(-1), dt.type(0)) / m       # d loss / d diff
        else:
            loss = np.sum(_softplus(-gamma * diff)) / m
            gd = -gamma * _sigmoid(-gamma * diff) / m
        G = -gd                                                       # through -S[i,j]
        G[np.arange(b), np.arange(b)] += gd.sum(axis=0)               # through S[j,j] (column j)
    else:
        wv = _f32_ratio(conf.neg_loss_weight, b - 1, dt)
        w = np.where(eye, dt.type(1), wv)                             # objectives.py:100-102
        if conf.loss == "skip-gram":
            label = np.where(eye, dt.type(1), dt.type(-1))            # :103
            loss = np.sum(w * _softplus(-label * S)) / dt.type(b)
            G = -w * label * _sigmoid(-label * S) / dt.type(b)
        else:
            label = np.where(eye, dt.type(1), dt.type(0))             # :110
            loss = np.sum(w * (S - label) ** 2) / dt.type(b)
            G = 2 * w * (S - label) / dt.type(b)
    return dt.type(loss), G.astype(dt)


def group_neg_shared_loss(S, pos_col, conf):
    """get_group_neg_shared_loss (objectives.py:163-220).  S: (b, n_u); pos_col[i] = cid_x[i].
    diff[i,j] = S[i,pos_i] - S[i,j]  (row-anchored, :175,196,200); num_negatives = n_u - 1 (:177)."""
    b, nu = S.shape
    dt = S.dtype
    lam, gamma = _f32_const(conf.neg_loss_weight, dt), _f32_const(conf.loss_gamma, dt)
    rows = np.arange(b)
    posmask = np.zeros((b, nu), dtype=bool)
    posmask[rows, pos_col] = True
    if conf.loss in ("max-margin", "log-loss"):
        diff = S[rows, pos_col][:, None] - S
        m = dt.type(b * nu)
        if conf.loss == "max-margin":
            act = gamma - diff                                        # :195-197 (no diag exemption)
            loss = np.sum(np.maximum(act, 0)) / m
            gd = np.where(act > 0, dt.type(-1), dt.type(0)) / m
        else:
            loss = np.sum(_softplus(-gamma * diff)) / m
            gd = -gamma * _sigmoid(-gamma * diff) / m
        G = -gd
        G[rows, pos_col] += gd.sum(axis=1)
    else:
        wv = lam / dt.type(nu - 1) if nu > 1 else dt.type(0)          # :203,211 (nu == 1: reference NaNs)
        w = np.where(posmask, dt.type(1), wv)
        if conf.loss == "skip-gram":
            label = np.where(posmask, dt.type(1), dt.type(-1))
            loss = np.sum(w * _softplus(-label * S)) / dt.type(b)
            G = -w * label * _sigmoid(-label * S) / dt.type(b)
        else:
            label = np.where(posmask, dt.type(1), dt.type(0))
            loss = np.sum(w * (S - label) ** 2) / dt.type(b)
            G = 2 * w * (S - label) / dt.type(b)
    return dt.type(loss), G.astype(dt)


def sampled_neg_shared_loss(S, conf):
    """get_sampled_neg_shared_loss (objectives.py:120-161).  S: (b, 1+k); column 0 positive."""
    b, k1 = S.shape
    k = k1 - 1
    dt = S.dtype
    lam, gamma = _f32_const(conf.neg_loss_weight, dt), _f32_const(conf.loss_gamma, dt)
    if conf.loss in ("max-margin", "log-loss"):
        diff = S[:, :1] - S[:, 1:]
        m = dt.type(b * k)
        if conf.loss == "max-margin":
            loss = np.sum(np.maximum(gamma - diff, 0)) / m
            gd = np.where(gamma - diff > 0, dt.type(-1), dt.type(0)) / m
        else:
            loss = np.sum(_softplus(-gamma * diff)) / m
            gd = -gamma * _sigmoid(-gamma * diff) / m
        G = np.concatenate([gd.sum(axis=1, keepdims=True), -gd], axis=1)
    else:
        wv = _f32_ratio(conf.neg_loss_weight, k, dt)
        w = np.ones((b, k1), dtype=dt)
        w[:, 1:] = wv
        label = np.ones((b, k1), dtype=dt)
        if conf.loss == "skip-gram":
            label[:, 1:] = -1
            loss = np.sum(w * _softplus(-label * S)) / dt.type(b)
            G = -w * label * _sigmoid(-label * S) / dt.type(b)
        else:
            label[:, 1:] = 0
            loss = np.sum(w * (S - label) ** 2) / dt.type(b)
            G = 2 * w * (S - label) / dt.type(b)
    return dt.type(loss), G.astype(dt)



def neg_shared_row_block(raw_u, raw_v, rows, conf, dtype=np.float32):
    """Rows `rows` of the b x b neg_shared step (pointwise losses) -- the share of one rank in the build's
    same-global-batch multi-GPU form (no reference counterpart; the whole matrix is get_neg_shared_loss,
    objectives.py:99-112, on InteractionDot 'matmul', interaction_dot.py:100-103).
    raw_u: (len(rows), d) raw user rows of the global positions `rows`; raw_v: (b, d) raw item rows of ALL positions.
    Returns (loss share incl. this block's part of the activity regulariser, dL/d raw_u (len(rows), d), the block's
    PARTIAL dL/d raw_v (b, d)); summed over a partition of the rows they are the one-device step's loss / gradients."""
    dt = np.dtype(dtype)
    raw_u, raw_v = np.asarray(raw_u, dtype=dt), np.asarray(raw_v, dtype=dt)
    rows = np.asarray(rows, dtype=np.int64)
    b = raw_v.shape[0]
    assert conf.loss in ("skip-gram", "mse") and b == conf.batch_size_p
    hat_u, inv_u = l2_normalize(raw_u) if conf.normalize_user else (raw_u, None)
    hat_v, inv_v = l2_normalize(raw_v) if conf.normalize_item else (raw_v, None)
    S = hat_u @ hat_v.T
    pos = np.zeros(S.shape, dtype=bool)
    pos[np.arange(len(rows)), rows] = True
    wv = _f32_ratio(conf.neg_loss_weight, b - 1, dt)
    w = np.where(pos, dt.type(1), wv)
    if conf.loss == "skip-gram":
        label = np.where(pos, dt.type(1), dt.type(-1))
        loss = np.sum(w * _softplus(-label * S)) / dt.type(b)
        G = -w * label * _sigmoid(-label * S) / dt.type(b)
    else:
        label = np.where(pos, dt.type(1), dt.type(0))
        loss = np.sum(w * (S - label) ** 2) / dt.type(b)
        G = 2 * w * (S - label) / dt.type(b)
    G = G.astype(dt)
    reg = dt.type(conf.u_reg) * np.sum(raw_u * raw_u) / dt.type(b)       # utilities.py:122-135: mean over the b rows
    g_u = G @ hat_v
    g_u = l2_normalize_backward(hat_u, inv_u, g_u) if inv_u is not None else g_u
    g_u = g_u + 2 * dt.type(conf.u_reg) * raw_u / dt.type(b)
    g_v = G.T @ hat_u
    g_v = l2_normalize_backward(hat_v, inv_v, g_v) if inv_v is not None else g_v
    return float(loss + reg), g_u.astype(dt), g_v.astype(dt)


# ----------------------------------------------------------------------------------------
# item text tower of model_choice = basic_embedding
# ----------------------------------------------------------------------------------------
def adam_lr_t(lr, beta1, beta2, t):
    """lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)  (utils/optimizer.py:109-111; Keras Adam [K-mem])."""
    return float(lr) * np.sqrt(1.0 - float(beta2) ** t) / (1.0 - float(beta1) ** t)


class AdamSlot(object):
    """First / second moments of one parameter array and the Adam step on it.
    rows=None: dense step (Keras Adam; utils/optimizer.py:99-106 _apply_dense).
    rows=index array: only those rows move (utils/optimizer.py:108-134 _apply_sparse on the
    de-duplicated, summed IndexedSlices)."""

    def __init__(self, like):
        self.m = np.zeros_like(like)
        self.v = np.zeros_like(like)

    def step(self, p, g, lr_t, conf, rows=None):
        dt = p.dtype.type
        b1, b2, eps, lr_t = dt(conf.beta1), dt(conf.beta2), dt(conf.epsilon), dt(lr_t)
        if rows is None:
            self.m[...] = b1 * self.m + (dt(1) - b1) * g
            self.v[...] = b2 * self.v + (dt(1) - b2) * g * g
            p -= lr_t * self.m / (np.sqrt(self.v) + eps)
        else:
            self.m[rows] = b1 * self.m[rows] + (dt(1) - b1) * g[rows]
            self.v[rows] = b2 * self.v[rows] + (dt(1) - b2) * g[rows] * g[rows]
            p[rows] -= lr_t * self.m[rows] / (np.sqrt(self.v[rows]) + eps)

    def step_as_written(self, p, idx, g_pos, lr_t, conf):
        """utils/optimizer.py:108-134 literally, for an IndexedSlices (indices idx, one gradient row g_pos[j] per
        lookup position) that still holds duplicate indices: m / v gathered (old values), scatter_update'd per
        position -- the last position of an index wins [K-mem: TF's CPU scatter_update applies updates in order] --
        and lr_t m / (sqrt(v) + eps), gathered AFTER the update, scatter_sub'ed once per position."""
        dt = p.dtype.type
        b1, b2, eps, lr_t = dt(conf.beta1), dt(conf.beta2), dt(conf.epsilon), dt(lr_t)
        idx = np.asarray(idx)
        m_new = b1 * self.m[idx] + (dt(1) - b1) * g_pos                 # m_scaled + m_scaled_g_values, per position
        v_new = b2 * self.v[idx] + (dt(1) - b2) * g_pos * g_pos
        for j in range(len(idx)):                                       # scatter_update, in position order
            self.m[idx[j]] = m_new[j]
            self.v[idx[j]] = v_new[j]
        upd = lr_t * self.m[idx] / (np.sqrt(self.v[idx]) + eps)         # m_tp, v_tp: gathered from the updated slots
        np.subtract.at(p, idx, upd.astype(p.dtype))                     # scatter_sub: once per occurrence


class TowerOracle(object):
    """MeanPool (modules/content/mean_pool.py:46-110) as get_model wires it for
    'basic_embedding' (model_framework.py:90-91,98): word Embedding -> AverageEmbeddings
    (divides by L always, :27-33) -> Dense(d) -> BatchNormalization (batch statistics over the
    n_u unique items of the batch, :90-95) -> ReLU -> Dropout(0).
    [K-mem]: Keras-1 BatchNormalization eps = 1e-3, momentum = 0.99, biased variance, the
    `running_std` slot holds the variance; Dense init glorot_uniform; Embedding init uniform(+-0.05).
    All parameters take plain SGD steps (the build's optimiser)."""

    def __init__(self, C, W, Wd, bd, gamma, beta, lr, dtype=np.float32, eps=1e-3, momentum=0.99):
        self.dt = np.dtype(dtype)
        self.C = np.asarray(C)
        self.W = np.array(W, dtype=self.dt)
        self.Wd = np.array(Wd, dtype=self.dt)
        self.bd = np.array(bd, dtype=self.dt)
        self.gamma = np.array(gamma, dtype=self.dt)
        self.beta = np.array(beta, dtype=self.dt)
        self.rm = np.zeros_like(self.beta)
        self.rv = np.ones_like(self.beta)
        self.lr = self.dt.type(lr)
        self.eps, self.mom = self.dt.type(eps), self.dt.type(momentum)
        self.pooling = "average"   # conf.pooling (mean_pool.py:68-76): 'average' | 'max'
        self.Cid = None            # conf.use_content_id (mean_pool.py:102-108): Emb_Cid table (item_count, d), or None
        self.v_reg = self.dt.type(0)   # activity_l2(v_reg) on the Emb_Cid output
        self.c_reg = self.dt.type(0)   # activity_l2(c_reg) on the word Embedding's output (mean_pool.py:56-59, cnn_model.py:45-48)
        self.dense_act = "relu"        # item_dense_transform['dense_hidden_actv'] (mean_pool.py:83,96): any of ACTIVATIONS
        self.dense_mask = None         # Dropout(dense_hidden_dropout) (mean_pool.py:84,97): (n, d) mask INCLUDING its 1/keep scale
        self.no_dense = False          # item_dense_transform falsy (mean_pool.py:78-101, else branch): the pooled block is the item row
        self.extra_loss = 0.0
        self.bag = None                # use_bag(): items x words count matrix (scipy CSR) -- the same sums, bag by bag
        self.no_bn = False             # conf.no_BN (mean_pool.py:88-93): Dense -> Activation -> Dropout, no BatchNormalization

    def use_bag(self):
        """Average pooling as a bag of words: sum_l W[tok[l]] * s[tok[l]] == sum_w count[item, w] * s[w] * W[w].  The same
        real-number sum as the token-wise form below taken word by word instead of position by position (~45 distinct
        words instead of 200 positions per item); in float64 the two agree to ~1e-15 relative
        (tests/test_oracle_step.py::test_bag_form_of_the_mean_pool_oracle).  For the epoch-long parity loops at the
        BASELINE shape, where the token-wise form's (n, L, wd) temporaries cost 0.2 s per step."""
        import scipy.sparse as sp
        n_items, L = self.C.shape
        rows = np.repeat(np.arange(n_items), L)
        self.bag = sp.csr_matrix((np.ones(rows.size, dtype=self.dt), (rows, self.C.reshape(-1))),
                                 shape=(n_items, self.W.shape[0]))
        self.bag.sum_duplicates()
        return self

    PARAMS = ("W", "Wd", "bd", "gamma", "beta")

    # Keras-1 activations by name (keras/activations.py; hard_sigmoid = clip(0.2 x + 0.5, 0, 1) [K-mem]): value, derivative
    ACTIVATIONS = {
        "relu": (lambda y: np.maximum(y, 0), lambda y: (y > 0).astype(y.dtype)),
        "tanh": (np.tanh, lambda y: 1 - np.tanh(y) ** 2),
        "sigmoid": (lambda y: 1 / (1 + np.exp(-y)), lambda y: (1 / (1 + np.exp(-y))) * (1 - 1 / (1 + np.exp(-y)))),
        "hard_sigmoid": (lambda y: np.clip(0.2 * y + 0.5, 0, 1), lambda y: 0.2 * ((y > -2.5) & (y < 2.5)).astype(y.dtype)),
        "softplus": (lambda y: np.logaddexp(y, 0), lambda y: 1 / (1 + np.exp(-y))),
        "softsign": (lambda y: y / (1 + np.abs(y)), lambda y: 1 / (1 + np.abs(y)) ** 2),
        "linear": (lambda y: y, lambda y: np.ones_like(y)),
    }

    def _front_forward(self, toks, row_scale, train):
        """tokens (n, L) -> pooled block h (n, hd): here the mean of the word rows."""
        if self.bag is not None and self.pooling == "average":
            import scipy.sparse as sp
            A = self.bag[self._cur_ids]
            if row_scale is not None:
                A = A @ sp.diags(np.asarray(row_scale, dtype=self.dt))
            if train:
                self.cache["A"] = A
            return np.asarray(A @ self.W, dtype=self.dt) / self.dt.type(toks.shape[1])
        E = self.W[toks]                                       # (n_u, L, wd)
        if row_scale is not None:
            E = E * row_scale[toks][..., None]
        if self.pooling == "max":                              # mean_pool.py:70-72: MaxPooling1D(max_content_len)
            am = E.argmax(axis=1)                              # (n, wd) position of each column's maximum
            if train:
                self.cache["argmax"] = am
            return np.take_along_axis(E, am[:, None, :], axis=1)[:, 0, :]
        return E.sum(axis=1) / self.dt.type(toks.shape[1])     # mean_pool.py:30: divide by L always

    def _front_backward(self, dh):
        """dL/dh -> gradients of the front's parameters (dict), from self.cache."""
        c = self.cache
        L = self.dt.type(c["toks"].shape[1])
        if "A" in c:                   # use_bag(): dW[w] = sum_items count * s[w] * dh[item] / L
            return dict(W=np.asarray(c["A"].T @ (dh / L), dtype=self.dt))
        dW = np.zeros_like(self.W)
        if self.pooling == "max":      # the gradient of column c goes to the word at the argmax position [K-mem: first maximum]
            am = c["argmax"]
            words = np.take_along_axis(c["toks"], am, axis=1)  # (n, wd) word that supplied each column's maximum
            g = dh if c["row_scale"] is None else dh * c["row_scale"][words]
            cols = np.broadcast_to(np.arange(dh.shape[1])[None, :], dh.shape)
            np.add.at(dW, (words, cols), g.astype(self.dt))
            return dict(W=dW)
        contrib = np.repeat(dh[:, None, :], c["toks"].shape[1], axis=1) / L
        if c["row_scale"] is not None:
            contrib = contrib * c["row_scale"][c["toks"]][..., None]
        np.add.at(dW, c["toks"], contrib.astype(self.dt))
        return dict(W=dW)

    def _plus_content_id(self, out, cid_u, train):
        """mean_pool.py:102-108: h + Emb_Cid[cid]; activity_l2(v_reg): v_reg * sum_d mean_n(row^2) joins the loss."""
        self.extra_loss = 0.0
        if self.Cid is None:
            return out
        rows = self.Cid[np.asarray(cid_u)]
        if train:
            self.cache["cid_u"] = np.asarray(cid_u)
            self.extra_loss = float(self.v_reg * np.sum(np.mean(rows * rows, axis=0)))
        return (out + rows).astype(self.dt)

    def _content_id_update(self, dA, conf, t):
        if self.Cid is None:
            return
        ids = self.cache["cid_u"]                              # unique ids of the block: one gradient row each
        n = self.dt.type(len(ids))
        g = (dA + 2 * self.v_reg * self.Cid[ids] / n).astype(self.dt)
        if getattr(self, "export_only", False):
            raise NotImplementedError("use_content_id with the data-parallel export")
        if conf is None or conf.optimizer == "sgd":
            self.Cid[ids] -= self.lr * g
        else:
            if not hasattr(self, "slot_cid"):
                self.slot_cid = AdamSlot(self.Cid)
            full = np.zeros_like(self.Cid)
            full[ids] = g
            self.slot_cid.step(self.Cid, full, adam_lr_t(self.lr, conf.beta1, conf.beta2, t), conf,
                               ids if conf.optimizer == "sparse_adam" else None)

    def forward(self, cid_u, train=True, row_scale=None):
        toks = self.C[np.asarray(cid_u)]                       # (n_u, L)
        self._cur_ids = np.asarray(cid_u)
        if train:
            self.cache = dict(toks=toks, row_scale=row_scale)
        h = self._front_forward(toks, row_scale, train)
        if self.no_dense:          # no Dense / BatchNorm / Activation: h itself (+ Emb_Cid, + c_reg below)
            if train:
                self.cache.update(h=h)
            out = self._plus_content_id(h.astype(self.dt), cid_u, train)
            if train and self.c_reg != 0:
                E = self.W[toks] if row_scale is None else self.W[toks] * row_scale[toks][..., None]
                self.extra_loss += float(self.c_reg * np.sum(np.mean(E * E, axis=0)))
            return out
        z = h @ self.Wd + self.bd
        if self.no_bn:     # mean_pool.py:88-93: `assert conf.no_BN == True` holds, BatchNormalization1 is never applied
            y = z
            if train:
                self.cache.update(h=h, y=y)
        else:
            if train:
                mu = z.mean(axis=0)
                var = ((z - mu) ** 2).mean(axis=0)
            else:
                mu, var = self.rm, self.rv
            inv = 1.0 / np.sqrt(var + self.eps)
            xhat = (z - mu) * inv
            y = self.gamma * xhat + self.beta
            if train:
                self.cache.update(h=h, xhat=xhat, inv=inv, y=y, mu=mu, var=var)
        if self.dense_act == "softmax":    # keras/activations.py softmax on a 2-D tensor: over the row's item_dim columns
            e = np.exp(y - y.max(axis=1, keepdims=True))
            act = e / e.sum(axis=1, keepdims=True)
            if train:
                self.cache["softmax"] = act
        else:
            act = self.ACTIVATIONS[self.dense_act][0](y)
        if train and self.dense_mask is not None:
            act = act * self.dense_mask[:len(act)]
        out = self._plus_content_id(act.astype(self.dt), cid_u, train)
        if train and self.c_reg != 0:
            # L1L2Act (utilities.py:122-135) on the Embedding output (n, L, wd), AFTER its row dropout [K-mem: Keras-1 Embedding
            # scales the table rows, the regulariser sees the gathered result]: c_reg * sum_{l, c} mean_n(E^2)
            E = self.W[toks] if row_scale is None else self.W[toks] * row_scale[toks][..., None]
            self.extra_loss += float(self.c_reg * np.sum(np.mean(E * E, axis=0)))
        return out

    def _c_reg_grad(self, dW):
        """d(c_reg * sum mean_n(E^2)) / dW: 2 c_reg / n * scale[w]^2 * W[w] per occurrence of w in the block."""
        if self.c_reg == 0:
            return
        c = self.cache
        toks, rs = c["toks"], c["row_scale"]
        n = self.dt.type(toks.shape[0])
        E = self.W[toks] if rs is None else self.W[toks] * (rs[toks] ** 2)[..., None]
        np.add.at(dW, toks, (2 * self.c_reg / n * E).astype(self.dt))

    def backward_update(self, dA, conf=None, t=None):
        """conf / t: OracleConf with optimizer != 'sgd' and the 1-based optimiser step."""
        c = self.cache
        self._content_id_update(dA, conf, t)
        if self.no_dense:          # dL/dh = dL/dE; only the front (word table, conv weights) has parameters
            self.grads = {k: np.zeros_like(getattr(self, k)) for k in ("Wd", "bd", "gamma", "beta")}
            self.grads.update(self._front_backward(np.asarray(dA, dtype=self.dt)))
            self._c_reg_grad(self.grads["W"])
            if getattr(self, "export_only", False):
                return
            live = [k for k in self.PARAMS if k not in ("Wd", "bd", "gamma", "beta")]   # W, and a single-bank conv front's weights
            if conf is None or conf.optimizer == "sgd":
                for k in live:
                    getattr(self, k).__isub__(self.lr * self.grads[k].astype(self.dt))
            else:
                if not hasattr(self, "slots"):
                    self.slots = {k: AdamSlot(getattr(self, k)) for k in live}
                lr_t = adam_lr_t(self.lr, conf.beta1, conf.beta2, t)
                rows = None
                if conf.optimizer == "sparse_adam":
                    toks = c["toks"].reshape(-1)
                    if c["row_scale"] is not None:
                        toks = toks[c["row_scale"][toks] != 0]
                    rows = np.unique(toks)
                for k in live:
                    self.slots[k].step(getattr(self, k), self.grads[k].astype(self.dt), lr_t, conf, rows if k == "W" else None)
            return
        n = self.dt.type(dA.shape[0])
        if self.dense_act == "softmax":    # dy_c = s_c (g_c - sum_j g_j s_j), g = dA * mask
            g = dA if self.dense_mask is None else dA * self.dense_mask[:len(dA)]
            sm = c["softmax"]
            dy = sm * (g - (g * sm).sum(axis=1, keepdims=True))
        else:
            dy = dA * self.ACTIVATIONS[self.dense_act][1](c["y"])
            if self.dense_mask is not None:
                dy = dy * self.dense_mask[:len(dy)]
        if self.no_bn:     # y = z: the Dense bias has a gradient of its own, gamma / beta none
            dgamma, dbeta, dz = np.zeros_like(self.gamma), np.zeros_like(self.beta), dy
        else:
            dgamma = (dy * c["xhat"]).sum(axis=0)
            dbeta = dy.sum(axis=0)
            dxhat = dy * self.gamma
            dz = c["inv"] / n * (n * dxhat - dxhat.sum(axis=0) - c["xhat"] * (dxhat * c["xhat"]).sum(axis=0))
        dWd = c["h"].T @ dz
        dbd = dz.sum(axis=0)
        dh = dz @ self.Wd.T
        self.grads = dict(Wd=dWd, bd=dbd, gamma=dgamma, beta=dbeta)
        self.grads.update(self._front_backward(dh.astype(self.dt)))
        dW = self.grads["W"]
        self._c_reg_grad(dW)
        if not self.no_bn:
            self.rm = self.mom * self.rm + (1 - self.mom) * c["mu"]
            self.rv = self.mom * self.rv + (1 - self.mom) * c["var"]
        if getattr(self, "export_only", False):   # data-parallel: gradients leave, nothing is applied
            return
        if conf is None or conf.optimizer == "sgd":
            for k in self.PARAMS:
                getattr(self, k).__isub__(self.lr * self.grads[k].astype(self.dt))
        else:
            if not hasattr(self, "slots"):
                self.slots = {k: AdamSlot(getattr(self, k)) for k in self.PARAMS}
            lr_t = adam_lr_t(self.lr, conf.beta1, conf.beta2, t)
            rows = None
            if conf.optimizer == "sparse_adam":   # words of the batch that were not dropped
                toks = c["toks"].reshape(-1)
                if c["row_scale"] is not None:
                    toks = toks[c["row_scale"][toks] != 0]
                rows = np.unique(toks)
            self.slots["W"].step(self.W, dW, lr_t, conf, rows)
            for k in self.PARAMS:
                if k != "W":
                    self.slots[k].step(getattr(self, k), self.grads[k].astype(self.dt), lr_t, conf)

    def flat_grads(self):
        """[W | Wd | bd, gamma, beta at 128-float pitch] -- the layout of nncf_tower_grad_floats."""
        g = self.grads
        pad = lambda v: np.concatenate([v, np.zeros(128 - len(v), dtype=self.dt)])  # noqa: E731
        return np.concatenate([g["W"].reshape(-1), g["Wd"].reshape(-1), pad(g["bd"]), pad(g["gamma"]),
                               pad(g["beta"])]).astype(self.dt)

    def apply_flat(self, flat, scale=1.0):
        """sparse-SGD step with scale * flat (after the data-parallel all-reduce)."""
        nW, nD, d = self.W.size, self.Wd.size, len(self.bd)
        f = np.asarray(flat, dtype=self.dt) * self.dt.type(scale)
        self.W -= self.lr * f[:nW].reshape(self.W.shape)
        self.Wd -= self.lr * f[nW:nW + nD].reshape(self.Wd.shape)
        self.bd -= self.lr * f[nW + nD:nW + nD + d]
        self.gamma -= self.lr * f[nW + nD + 128:nW + nD + 128 + d]
        self.beta -= self.lr * f[nW + nD + 256:nW + nD + 256 + d]

class CnnTowerOracle(TowerOracle):
    """CNN item tower of model_choice = cnn_embedding (modules/content/cnn_model.py:11-145) for one
    convolution layer with a single filter length (the 'best' preset: filter_lengths [3],
    cnn_embedding_conf.py:129): word Embedding (+ row dropout) -> Convolution1D(num_filters,
    filter_length, border_mode 'same', linear, :82-88) -> BatchNormalization over (items, positions)
    per filter (:89-95) -> conv_activation relu -> Dropout(conv_dropout_rate) (:96-97) ->
    AveragePooling1D over the whole length (:100-104) -> Flatten -> Dense(item_dim) -> BatchNorm
    -> ReLU (:112-126), the same tail as MeanPool.
    Wc: (fl, wd, F) as Keras stores it (fl, 1, wd, F); conv_mask(n, L, F): the dropout mask
    INCLUDING its 1/keep scale for the next train-mode forward (None = no dropout).
    [K-mem] TF 'SAME' padding for odd fl: (fl-1)/2 zeros on each side."""

    PARAMS = ("W", "Wc", "bc", "g1", "b1", "Wd", "bd", "gamma", "beta")

    def __init__(self, C, W, Wc, bc, g1, b1, Wd, bd, gamma, beta, lr, dtype=np.float32, eps=1e-3, momentum=0.99):
        TowerOracle.__init__(self, C, W, Wd, bd, gamma, beta, lr, dtype=dtype, eps=eps, momentum=momentum)
        self.Wc = np.array(Wc, dtype=self.dt)
        self.bc = np.array(bc, dtype=self.dt)
        self.g1 = np.array(g1, dtype=self.dt)
        self.b1 = np.array(b1, dtype=self.dt)
        self.rm1 = np.zeros_like(self.b1)
        self.rv1 = np.ones_like(self.b1)
        self.conv_mask = None
        assert self.Wc.shape[0] % 2 == 1, "odd filter lengths only"

    def flat_grads(self):
        """TowerOracle.flat_grads + [Wc | bc, g1, b1 at 128-float pitch] (nncf_tower_grad_floats)."""
        g = self.grads
        pad = lambda v: np.concatenate([v, np.zeros(128 - len(v), dtype=self.dt)])  # noqa: E731
        return np.concatenate([TowerOracle.flat_grads(self), g["Wc"].reshape(-1), pad(g["bc"]), pad(g["g1"]),
                               pad(g["b1"])]).astype(self.dt)

    def apply_flat(self, flat, scale=1.0):
        TowerOracle.apply_flat(self, flat, scale)
        off = self.W.size + self.Wd.size + 384
        nC, F = self.Wc.size, len(self.bc)
        f = np.asarray(flat, dtype=self.dt) * self.dt.type(scale)
        self.Wc -= self.lr * f[off:off + nC].reshape(self.Wc.shape)
        self.bc -= self.lr * f[off + nC:off + nC + F]
        self.g1 -= self.lr * f[off + nC + 128:off + nC + 128 + F]
        self.b1 -= self.lr * f[off + nC + 256:off + nC + 256 + F]

    def _front_forward(self, toks, row_scale, train):
        fl, pad = self.Wc.shape[0], self.Wc.shape[0] // 2
        n, L = toks.shape
        X = self.W[toks]
        if row_scale is not None:
            X = X * row_scale[toks][..., None]
        Xp = np.zeros((n, L + 2 * pad, X.shape[2]), dtype=self.dt)
        Xp[:, pad:pad + L] = X
        Y = np.zeros((n, L, self.Wc.shape[2]), dtype=self.dt) + self.bc
        for tap in range(fl):
            Y += Xp[:, tap:tap + L] @ self.Wc[tap]
        if train:
            mu = Y.mean(axis=(0, 1))
            var = ((Y - mu) ** 2).mean(axis=(0, 1))
        else:
            mu, var = self.rm1, self.rv1
        inv = 1.0 / np.sqrt(var + self.eps)
        xh = (Y - mu) * inv
        y1 = self.g1 * xh + self.b1
        a = np.maximum(y1, 0)
        mask = self.conv_mask[:n] if (train and self.conv_mask is not None) else None   # rows = block slots
        if mask is not None:
            a = a * mask
        if getattr(self, "conv_pooling", "average") == "max":   # cnn_model.py:107-109: MaxPooling1D over the whole length
            amax = a.argmax(axis=1)                              # [K-mem] first maximum in position order
            if train:
                self.cache.update(Xp=Xp, xh1=xh, inv1=inv, y1=y1, mask=mask, mu1=mu, var1=var, amax=amax)
            return np.take_along_axis(a, amax[:, None, :], axis=1)[:, 0, :].astype(self.dt)
        if train:
            self.cache.update(Xp=Xp, xh1=xh, inv1=inv, y1=y1, mask=mask, mu1=mu, var1=var)
        return (a.sum(axis=1) / self.dt.type(L)).astype(self.dt)

    def _front_backward(self, dh):
        c = self.cache
        fl, pad = self.Wc.shape[0], self.Wc.shape[0] // 2
        n, Lp, wd = c["Xp"].shape
        L = Lp - 2 * pad
        if getattr(self, "conv_pooling", "average") == "max":   # the gradient goes to the position that supplied the maximum
            da = np.zeros((n, L, dh.shape[1]), dtype=self.dt)
            np.put_along_axis(da, c["amax"][:, None, :], dh[:, None, :].astype(self.dt), axis=1)
        else:
            da = np.repeat(dh[:, None, :], L, axis=1) / self.dt.type(L)
        if c["mask"] is not None:
            da = da * c["mask"]
        dy1 = da * (c["y1"] > 0)
        N = self.dt.type(n * L)
        dg1 = (dy1 * c["xh1"]).sum(axis=(0, 1))
        db1 = dy1.sum(axis=(0, 1))
        dxh = dy1 * self.g1
        dY = c["inv1"] / N * (N * dxh - dxh.sum(axis=(0, 1)) - c["xh1"] * (dxh * c["xh1"]).sum(axis=(0, 1)))
        dY = dY.astype(self.dt)
        dWc = np.zeros_like(self.Wc)
        dXp = np.zeros_like(c["Xp"])
        for tap in range(fl):
            dWc[tap] = np.einsum("nlk,nlf->kf", c["Xp"][:, tap:tap + L], dY)
            dXp[:, tap:tap + L] += dY @ self.Wc[tap].T
        dX = dXp[:, pad:pad + L]
        if c["row_scale"] is not None:
            dX = dX * c["row_scale"][c["toks"]][..., None]
        dW = np.zeros_like(self.W)
        np.add.at(dW, c["toks"], dX.astype(self.dt))
        self.rm1 = self.mom * self.rm1 + (1 - self.mom) * c["mu1"]
        self.rv1 = self.mom * self.rv1 + (1 - self.mom) * c["var1"]
        return dict(W=dW, Wc=dWc, bc=dY.sum(axis=(0, 1)), g1=dg1, b1=db1)


class CnnBankTowerOracle(TowerOracle):
    """CNN item tower with a FILTER BANK (modules/content/cnn_model.py:63-81, the `default` preset's
    filter_lengths [[1, 3, 5]], cnn_embedding_conf.py:42-45): one Convolution1D('same', linear) per listed length
    over the same word rows, outputs merged with mode 'concat' along the filter axis (:80), then ONE
    BatchNormalization over (items, positions) per channel of the concatenation (:89-95) -- per-channel statistics, so
    it factors per bank --, relu, Dropout, AveragePooling1D over the whole length, Flatten, Dense, BatchNorm, ReLU.
    Wc / bc / g1 / b1: lists with one entry per filter length ((fl, wd, F), (F,), (F,), (F,)); conv_mask: list of
    (n, L, F) masks including their 1/keep scale, or None.  With one bank this is CnnTowerOracle
    (tests/test_oracle_step.py checks that, and the bank against one zero-padded convolution of the longest length)."""

    def __init__(self, C, W, Wc, bc, g1, b1, Wd, bd, gamma, beta, lr, dtype=np.float32, eps=1e-3, momentum=0.99):
        TowerOracle.__init__(self, C, W, Wd, bd, gamma, beta, lr, dtype=dtype, eps=eps, momentum=momentum)
        self.Wc = [np.array(v, dtype=self.dt) for v in Wc]
        self.bc = [np.array(v, dtype=self.dt) for v in bc]
        self.g1 = [np.array(v, dtype=self.dt) for v in g1]
        self.b1 = [np.array(v, dtype=self.dt) for v in b1]
        self.rm1 = [np.zeros_like(v) for v in self.b1]
        self.rv1 = [np.ones_like(v) for v in self.b1]
        self.conv_mask = None
        self.nb = len(self.Wc)
        assert all(w.shape[0] % 2 == 1 for w in self.Wc), "odd filter lengths only"
        assert self.Wd.shape[0] == sum(w.shape[2] for w in self.Wc), "Dense input = all banks' filters"

    PARAMS = ("W", "Wd", "bd", "gamma", "beta")      # + the per-bank lists, stepped in backward_update below

    def _front_forward(self, toks, row_scale, train):
        n, L = toks.shape
        X = self.W[toks]
        if row_scale is not None:
            X = X * row_scale[toks][..., None]
        hs, banks = [], []
        for g in range(self.nb):
            Wc = self.Wc[g]
            fl, pad = Wc.shape[0], Wc.shape[0] // 2
            Xp = np.zeros((n, L + 2 * pad, X.shape[2]), dtype=self.dt)
            Xp[:, pad:pad + L] = X
            Y = np.zeros((n, L, Wc.shape[2]), dtype=self.dt) + self.bc[g]
            for tap in range(fl):
                Y += Xp[:, tap:tap + L] @ Wc[tap]
            if train:
                mu = Y.mean(axis=(0, 1))
                var = ((Y - mu) ** 2).mean(axis=(0, 1))
            else:
                mu, var = self.rm1[g], self.rv1[g]
            inv = 1.0 / np.sqrt(var + self.eps)
            xh = (Y - mu) * inv
            y1 = self.g1[g] * xh + self.b1[g]
            a = np.maximum(y1, 0)
            mask = self.conv_mask[g][:n] if (train and self.conv_mask is not None) else None
            if mask is not None:
                a = a * mask
            bk = dict(Xp=Xp, xh1=xh, inv1=inv, y1=y1, mask=mask, mu1=mu, var1=var)
            if getattr(self, "conv_pooling", "average") == "max":
                bk["amax"] = a.argmax(axis=1)
                hs.append(np.take_along_axis(a, bk["amax"][:, None, :], axis=1)[:, 0, :])
            else:
                hs.append(a.sum(axis=1) / self.dt.type(L))
            banks.append(bk)
        if train:
            self.cache["banks"] = banks
        return np.concatenate(hs, axis=1).astype(self.dt)       # Merge(mode='concat', concat_axis=-1)

    def _front_backward(self, dh):
        c = self.cache
        toks = c["toks"]
        n, L = toks.shape
        dX = np.zeros((n, L, self.W.shape[1]), dtype=self.dt)
        self.bank_grads = []
        off = 0
        for g in range(self.nb):
            b, Wc = c["banks"][g], self.Wc[g]
            fl, pad, F = Wc.shape[0], Wc.shape[0] // 2, Wc.shape[2]
            if getattr(self, "conv_pooling", "average") == "max":
                da = np.zeros((n, L, F), dtype=self.dt)
                np.put_along_axis(da, b["amax"][:, None, :], dh[:, None, off:off + F].astype(self.dt), axis=1)
            else:
                da = np.repeat(dh[:, None, off:off + F], L, axis=1) / self.dt.type(L)
            off += F
            if b["mask"] is not None:
                da = da * b["mask"]
            dy1 = da * (b["y1"] > 0)
            N = self.dt.type(n * L)
            dg1 = (dy1 * b["xh1"]).sum(axis=(0, 1))
            db1 = dy1.sum(axis=(0, 1))
            dxh = dy1 * self.g1[g]
            dY = b["inv1"] / N * (N * dxh - dxh.sum(axis=(0, 1)) - b["xh1"] * (dxh * b["xh1"]).sum(axis=(0, 1)))
            dY = dY.astype(self.dt)
            dWc = np.zeros_like(Wc)
            dXp = np.zeros_like(b["Xp"])
            for tap in range(fl):
                dWc[tap] = np.einsum("nlk,nlf->kf", b["Xp"][:, tap:tap + L], dY)
                dXp[:, tap:tap + L] += dY @ Wc[tap].T
            dX += dXp[:, pad:pad + L]
            self.bank_grads.append(dict(Wc=dWc, bc=dY.sum(axis=(0, 1)), g1=dg1, b1=db1))
            self.rm1[g] = self.mom * self.rm1[g] + (1 - self.mom) * b["mu1"]
            self.rv1[g] = self.mom * self.rv1[g] + (1 - self.mom) * b["var1"]
        if c["row_scale"] is not None:
            dX = dX * c["row_scale"][toks][..., None]
        dW = np.zeros_like(self.W)
        np.add.at(dW, toks, dX.astype(self.dt))
        return dict(W=dW)

    def backward_update(self, dA, conf=None, t=None):
        TowerOracle.backward_update(self, dA, conf, t)
        if getattr(self, "export_only", False):
            return
        for g in range(self.nb):
            for k in ("Wc", "bc", "g1", "b1"):
                p, grad = getattr(self, k)[g], self.bank_grads[g][k].astype(self.dt)
                if conf is None or conf.optimizer == "sgd":
                    p -= self.lr * grad
                else:
                    key = "%s%d" % (k, g)
                    if key not in self.slots:
                        self.slots[key] = AdamSlot(p)
                    self.slots[key].step(p, grad, adam_lr_t(self.lr, conf.beta1, conf.beta2, t), conf)


# ----------------------------------------------------------------------------------------
# the model: id-table towers ('mf', model_framework.py:59-63,85-88) + sparse SGD
# ----------------------------------------------------------------------------------------
class StepOracle(object):
    """Holds copies of the user / item tables and applies one reference step at a time.

    U: (user_count, d), V: (item_count, d) float arrays (copied, cast to `dtype`)."""

    def __init__(self, U, V, conf, dtype=np.float32, tower=None):
        self.dt = np.dtype(dtype)
        self.U = np.array(U, dtype=self.dt, copy=True)
        self.V = np.array(V, dtype=self.dt, copy=True) if V is not None else None
        self.conf = conf
        self.tower = tower  # TowerOracle: item rows come from the text tower (model_framework.py:98-106)
        self.t = 0  # optimiser steps taken (Keras `iterations`)
        self.slot_u = self.slot_v = None
        ib = getattr(conf, "interaction_bias", None)
        self.ub = np.zeros(self.U.shape[0], dtype=self.dt) if ib in ("user", "both") else None
        self.cb = None   # sized on first use (item count: the table, or set_item_count for tower models)
        self._want_cb = ib in ("item", "both")
        self.slot_ub = self.slot_cb = None
        if self._want_cb and self.V is not None:
            self.cb = np.zeros(self.V.shape[0], dtype=self.dt)
        self.row_scale = None  # per-word scale of Embedding row dropout for the next step (mean_pool.py:56-59)

    def _tower_in(self, cid):
        """Content models run on tf.unique(cid) and are re-gathered (model_framework.py:45-48,
        98-106): expose the compact block as this step's item 'table'."""
        self._gid = None          # global item id of every row of the item block (bias lookups)
        if self.tower is None:
            return cid
        cid_u, cid_x = tf_unique(cid)
        self._gid = cid_u
        self.V = self.tower.forward(cid_u, train=True, row_scale=self.row_scale)
        self._n_u = len(cid_u)
        return cid_x.astype(np.int64)

    def _tower_out(self, update):
        if self.tower is not None and update:
            self.tower.backward_update(self.last["dV"][:self._n_u], self.conf, self.t)

    # -- shared pieces -------------------------------------------------------------------
    def _user_side(self, uid):
        """Emb_U gather (+ activity_l2(u_reg) on the raw output, model_framework.py:59-61;
        utilities.py:122-135: u_reg * sum_d mean_n(E[n,d]^2)) then optional l2_normalize (:62-63)."""
        raw = self.U[uid]
        reg = self.dt.type(self.conf.u_reg) * np.sum(np.mean(raw * raw, axis=0))
        if self.conf.normalize_user:
            hat, inv = l2_normalize(raw)
        else:
            hat, inv = raw, None
        return raw, hat, inv, self.dt.type(reg)

    def _item_side(self, cid):
        raw = self.V[cid]
        if self.conf.normalize_item:
            hat, inv = l2_normalize(raw)
        else:
            hat, inv = raw, None
        return raw, hat, inv

    def _bias(self, uid, cid):
        """(ubias[uid], cbias[global id of item-block row cid]) or zeros (interaction_dot.py:96-99,104-107)."""
        bu = self.ub[uid] if self.ub is not None else np.zeros(len(uid), dtype=self.dt)
        if self._want_cb and self.cb is None:
            raise ValueError("item bias with a text tower: set oracle.cb = zeros(item_count) first")
        gid = self._gid[cid] if self._gid is not None else cid
        bc = self.cb[gid] if self.cb is not None else np.zeros(len(cid), dtype=self.dt)
        return bu.astype(self.dt), bc.astype(self.dt)

    def _apply_bias(self, uid, cid, gb_u, gb_v, lr_t):
        """gb_u[i] / gb_v[j]: dL/d(score) summed over the scores user row i / item row j takes part in."""
        opt = getattr(self.conf, "optimizer", "sgd")
        lr = self.dt.type(self.conf.learn_rate)
        for tab, ids, gb, name in ((self.ub, uid, gb_u, "slot_ub"),
                                   (self.cb, (self._gid[cid] if self._gid is not None else cid), gb_v, "slot_cb")):
            if tab is None:
                continue
            d = np.zeros_like(tab)
            np.add.at(d, ids, gb.astype(self.dt))
            self.last_gb = dict(getattr(self, "last_gb", {}), **{name[5:]: d})   # {'ub': ..., 'cb': ...}: summed per id (the sharded checkers)
            touched = np.unique(ids)
            if opt == "sgd":
                tab[touched] -= lr * d[touched]
            else:
                if getattr(self, name) is None:
                    setattr(self, name, AdamSlot(tab))
                getattr(self, name).step(tab, d, lr_t, self.conf, touched if opt == "sparse_adam" else None)

    def _apply(self, uid, cid, raw_u, hat_u, inv_u, hat_v, inv_v, g_uhat, g_vhat, gb_u=None, gb_v=None):
        """Back through the normalisation, add the regulariser gradient, scatter-SGD."""
        lr = self.dt.type(self.conf.learn_rate)
        n = self.dt.type(len(uid))
        g_u = l2_normalize_backward(hat_u, inv_u, g_uhat) if inv_u is not None else g_uhat
        g_u = g_u + 2 * self.dt.type(self.conf.u_reg) * raw_u / n
        g_v = l2_normalize_backward(hat_v, inv_v, g_vhat) if inv_v is not None else g_vhat
        dU = np.zeros_like(self.U)
        dV = np.zeros_like(self.V)
        np.add.at(dU, uid, g_u.astype(self.dt))
        np.add.at(dV, cid, g_v.astype(self.dt))
        tu, tv = np.unique(uid), np.unique(cid)
        opt = getattr(self.conf, "optimizer", "sgd")
        if opt == "sgd":
            self.U[tu] -= lr * dU[tu]
            if self.tower is None:
                self.V[tv] -= lr * dV[tv]
        else:
            self.t += 1
            lr_t = adam_lr_t(self.conf.learn_rate, self.conf.beta1, self.conf.beta2, self.t)
            if self.slot_u is None:
                self.slot_u = AdamSlot(self.U)
                self.slot_v = AdamSlot(self.V) if self.tower is None else None
            sparse = opt == "sparse_adam"
            if sparse and getattr(self.conf, "sparse_adam_duplicates", "sum") == "as_written":
                self.slot_u.step_as_written(self.U, uid, g_u.astype(self.dt), lr_t, self.conf)
                if self.tower is None:
                    self.slot_v.step_as_written(self.V, cid, g_v.astype(self.dt), lr_t, self.conf)
            else:
                self.slot_u.step(self.U, dU, lr_t, self.conf, tu if sparse else None)
                if self.tower is None:
                    self.slot_v.step(self.V, dV, lr_t, self.conf, tv if sparse else None)
        if gb_u is not None:
            self._apply_bias(uid, cid, gb_u, gb_v, adam_lr_t(self.conf.learn_rate, self.conf.beta1, self.conf.beta2, max(self.t, 1)))
        self.last = dict(g_u=g_u, g_v=g_v, dU=dU, dV=dV)

    # -- train_scheme = original (train_original.py:49-58; model 'model', :123,147-149) ---
    def step_original(self, uid, cid, y, update=True):
        uid, cid = np.asarray(uid, dtype=np.int64), np.asarray(cid, dtype=np.int64)
        cid = self._tower_in(cid)
        y = np.asarray(y).astype(self.dt).reshape(-1)
        raw_u, hat_u, inv_u, reg = self._user_side(uid)
        raw_v, hat_v, inv_v = self._item_side(cid)
        bu, bc = self._bias(uid, cid)
        s = np.sum(hat_u * hat_v, axis=1) + bu + bc                    # interaction_dot.py:92-99
        loss, g = original_loss(s, y, self.conf)
        self.scores = s
        if update:
            self._apply(uid, cid, raw_u, hat_u, inv_u, hat_v, inv_v, g[:, None] * hat_v, g[:, None] * hat_u, g, g)
        self._tower_out(update)
        return float(loss + reg) + float(getattr(self.tower, 'extra_loss', 0.0) or 0.0)   # + activity_l2(v_reg) of Emb_Cid

    # -- train_scheme = neg_shared (train_neg_shared.py:46-51; :128-129,153-155) ----------
    def step_neg_shared(self, uid, cid, update=True):
        uid, cid = np.asarray(uid, dtype=np.int64), np.asarray(cid, dtype=np.int64)
        cid = self._tower_in(cid)
        raw_u, hat_u, inv_u, reg = self._user_side(uid)
        raw_v, hat_v, inv_v = self._item_side(cid)
        bu, bc = self._bias(uid, cid)
        S = hat_u @ hat_v.T + bu[:, None] + bc[None, :]                # interaction_dot.py:100-107
        loss, G = neg_shared_loss(S, self.conf)
        self.scores = S
        if update:
            self._apply(uid, cid, raw_u, hat_u, inv_u, hat_v, inv_v, G @ hat_v, G.T @ hat_u, G.sum(axis=1), G.sum(axis=0))
        self._tower_out(update)
        return float(loss + reg) + float(getattr(self.tower, 'extra_loss', 0.0) or 0.0)   # + activity_l2(v_reg) of Emb_Cid

    # -- train_scheme = group_neg_shared (train_group_neg_shared.py:53-61; :131-136,158-161)
    def step_group_neg_shared(self, uid, cid, update=True):
        uid, cid = np.asarray(uid, dtype=np.int64), np.asarray(cid, dtype=np.int64)
        assert len(uid) == self.conf.batch_size_p                      # row_cidx_prefx, :38-39,133-134
        cid = self._tower_in(cid)
        cid_u, cid_x = tf_unique(cid)                                  # :45-48
        raw_u, hat_u, inv_u, reg = self._user_side(uid)
        raw_v, hat_v, inv_v = self._item_side(cid_u)                   # compact item block
        bu, bc = self._bias(uid, cid_u)
        S = hat_u @ hat_v.T + bu[:, None] + bc[None, :]                # (b, n_u)
        loss, G = group_neg_shared_loss(S, cid_x, self.conf)
        self.scores, self.cid_u, self.cid_x = S, cid_u, cid_x
        if update:
            self._apply(uid, cid_u, raw_u, hat_u, inv_u, hat_v, inv_v, G @ hat_v, G.T @ hat_u, G.sum(axis=1), G.sum(axis=0))
        self._tower_out(update)
        return float(loss + reg) + float(getattr(self.tower, 'extra_loss', 0.0) or 0.0)   # + activity_l2(v_reg) of Emb_Cid

    # -- train_scheme = sampled_neg_shared (train_sampled_neg_shared.py:46-52; :138-143,166-170)
    def step_sampled_neg_shared(self, uid, cid, update=True):
        b = self.conf.batch_size_p
        uid, cid = np.asarray(uid, dtype=np.int64), np.asarray(cid, dtype=np.int64)
        # the head uses U_emb_front = Emb_U(uid[:b]) WITHOUT normalisation (:64-65) and C_emb
        # sliced front/back (:114-117); the regulariser node is the front call.
        ufront = uid[:b]
        cid = self._tower_in(cid)
        raw_u = self.U[ufront]
        reg = self.dt.type(self.conf.u_reg) * np.sum(np.mean(raw_u * raw_u, axis=0))
        raw_v, hat_v, inv_v = self._item_side(cid)
        bu, bc = self._bias(ufront, cid)                                # interaction_dot.py:96-99,104-107
        pos = np.sum(raw_u * hat_v[:b], axis=1, keepdims=True) + (bu + bc[:b])[:, None]
        neg = raw_u @ hat_v[b:].T + bu[:, None] + bc[b:][None, :]
        S = np.concatenate([pos, neg], axis=1)
        loss, G = sampled_neg_shared_loss(S, self.conf)
        self.scores = S
        if update:
            g_u = G[:, :1] * hat_v[:b] + G[:, 1:] @ hat_v[b:]
            g_v = np.concatenate([G[:, :1] * raw_u, G[:, 1:].T @ raw_u], axis=0)
            self._apply(ufront, cid, raw_u, raw_u, None, hat_v, inv_v, g_u, g_v, G.sum(axis=1),
                        np.concatenate([G[:, 0], G[:, 1:].sum(axis=0)]))
        self._tower_out(update)
        return float(loss + reg) + float(getattr(self.tower, 'extra_loss', 0.0) or 0.0)   # + activity_l2(v_reg) of Emb_Cid

    # -- inference heads (model_framework.py:153,173-176) --------------------------------
    def predict_neg_shared(self, uid, cid):
        uid, cid = np.asarray(uid, dtype=np.int64), np.asarray(cid, dtype=np.int64)
        _, hat_u, _, _ = self._user_side(uid)
        _, hat_v, _ = self._item_side(cid)
        self._gid = None
        bu, bc = self._bias(uid, cid)
        return hat_u @ hat_v.T + bu[:, None] + bc[None, :]


# ----------------------------------------------------------------------------------------
# batch assembly as the trainers do it
# ----------------------------------------------------------------------------------------
def assemble_original(batch_p, negatives, loss):
    """train_original.py:49-56: positives, then each positive repeated k times consecutively
    with the item replaced by a sampled negative; label -1 (skip-gram) or 0 (:13-14)."""
    batch_p = np.asarray(batch_p)
    b = batch_p.shape[0]
    k = len(negatives) // b
    neg_sign = -1 if loss == "skip-gram" else 0
    batch_n = batch_p.repeat(k, axis=0)
    batch_n[:, 1] = negatives
    batch_n[:, 2] = neg_sign
    tb = np.vstack((batch_p, batch_n))
    return tb[:, 0], tb[:, 1], tb[:, 2]


# ----------------------------------------------------------------------------------------
# evaluation: whole@k  (objectives.py:296-370,395-421; metrics_ranking.py:6-35)
# ----------------------------------------------------------------------------------------
def eval_multiple(true_scores, pred_scores, topk, noise=None):
    """metrics_ranking.py:6-35 with the tie-break noise passed in (default: none => ties are
    broken by the true label, then arbitrarily -- use tie-free scores in tests)."""
    pred_scores = np.asarray(pred_scores)
    true_scores = np.asarray(true_scores)
    idx = np.argpartition(-pred_scores, topk)[:topk]
    if noise is None:
        noise = np.zeros(topk)
    rec = sorted(zip(pred_scores[idx], noise, true_scores[idx]), reverse=True)
    nhits_topk, sumap, nh = 0., 0., 0.
    k = topk
    for i in range(len(rec)):
        if rec[i][-1] != 0.:
            nh += 1.0
            if i < k:
                nhits_topk += 1
                sumap += nh / (i + 1.0)
    nhits = np.sum(true_scores)
    if nhits != 0:
        return sumap / min(nhits, k), nhits_topk / nhits, nhits_topk / k
    return 0., 0., 0.


def evaluate_whole(U_hat, V_hat, links, candidate_items, topk):
    """Evaluator whole@k for one candidate set (objectives.py:395-421,296-321,333-370):
    truth[user, slot(item)] = 1 for every link whose item is a candidate; users with no
    truth are skipped; returns mean (map, recall, precision)."""
    cand = np.asarray(sorted(set(candidate_items)))
    slot = {it: i for i, it in enumerate(cand)}
    nu = U_hat.shape[0]
    truth = np.zeros((nu, len(cand)), dtype=np.int32)
    for u, i in links:
        if i in slot:
            truth[u, slot[i]] = 1
    pred = U_hat @ V_hat[cand].T
    keep = truth.sum(axis=1) > 0
    res = [eval_multiple(truth[u], pred[u], topk) for u in np.nonzero(keep)[0]]
    res = np.asarray(res, dtype=np.float64)
    return dict(map=res[:, 0].mean(), recall=res[:, 1].mean(), precision=res[:, 2].mean(), users=int(keep.sum()))
