"""Synthetic "CiteULike-a-shaped" data in the reference's pickle schema (data/readme.txt:1-9).

The real blob (data/citeulike.7z) is not shipped with the reference (.MISSING_LARGE_BLOBS),
so every measurement runs on data of the same shape: 16,000 users x 17,000 items,
200,000 distinct (user, item, 1) training links, a cold-item split (20 % of the items and
their links form `test`), Zipf item popularity, log-normal user activity, and planted
low-rank preferences so that Recall@k is learnable.  SURVEY.md section 8(d), input A.
"""
import numpy as np


class DataSpec(object):
    """configs/data_utils.py:27-35"""

    def __init__(self, user_count, word_count, item_count, max_content_len):
        self.user_count = user_count
        self.word_count = word_count
        self.item_count = item_count
        self.max_content_len = max_content_len
        self.W_pretrain = None
        self.C_pretrain = None


def _sample_links(rs, user_w, item_ids, item_w, P, Q, n_links, beta, chunk=2000):
    """n_links distinct (u, i) pairs, P(u,i) ~ user_w[u] * item_w[i] * exp(beta * P_u.Q_i)."""
    U = user_w.size
    deg = np.maximum(1, np.round(user_w / user_w.sum() * n_links)).astype(np.int64)
    deg = np.minimum(deg, item_ids.size)
    # fix the total exactly
    diff = int(n_links - deg.sum())
    order = np.argsort(-user_w)
    idx = 0
    while diff != 0:
        u = order[idx % U]
        if diff > 0 and deg[u] < item_ids.size:
            deg[u] += 1
            diff -= 1
        elif diff < 0 and deg[u] > 1:
            deg[u] -= 1
            diff += 1
        idx += 1
    rows = []
    logw = np.log(item_w)
    for u0 in range(0, U, chunk):
        u1 = min(U, u0 + chunk)
        score = beta * (P[u0:u1] @ Q[item_ids].T) + logw[None, :]
        # Gumbel top-k == sampling without replacement proportional to exp(score)
        g = score + rs.gumbel(size=score.shape)
        for r, u in enumerate(range(u0, u1)):
            k = deg[u]
            top = np.argpartition(-g[r], k - 1)[:k]
            rows.append(np.stack([np.full(k, u, dtype=np.int64), item_ids[top]], 1))
    return np.concatenate(rows, 0)


def make_citeulike_like(users=16000, items=17000, train_links=200000, test_frac=0.2, rank=16, beta=3.0,
                        seed=20170813, content_len=0, vocab=8001, text_scale=0.65):
    """Returns a dict with the keys of data_split_cold_item.pkl (data/readme.txt:3):
    'C', 'test_items', 'train_items', 'test', 'test_seen', 'train' (int64 (n,3) triplets)."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(items)
    item_w = np.empty(items)
    item_w[perm] = 1.0 / np.arange(1, items + 1)            # Zipf(1.0) over a random permutation
    user_w = np.exp(rs.normal(0.0, 1.0, users))             # log-normal(sigma = 1) activity
    P = rs.normal(0, 1.0 / np.sqrt(rank), (users, rank))
    Q = rs.normal(0, 1.0, (items, rank))
    n_test_items = int(round(items * test_frac))
    shuffled = rs.permutation(items)
    test_items = np.sort(shuffled[:n_test_items])
    train_items = np.sort(shuffled[n_test_items:])

    train = _sample_links(rs, user_w, train_items, item_w[train_items], P, Q, train_links, beta)
    # every train item needs >= 1 link (the sampler's degree table is built from train)
    have = np.zeros(items, dtype=bool)
    have[train[:, 1]] = True
    missing = train_items[~have[train_items]]
    if missing.size:
        # re-point links of over-represented (user, item) pairs: take links whose item keeps >= 2
        cnt = np.bincount(train[:, 1], minlength=items)
        ucnt = np.bincount(train[:, 0], minlength=users)
        cand = rs.permutation(train.shape[0])
        j = 0
        for it in missing:
            while True:
                r = cand[j]
                j += 1
                if cnt[train[r, 1]] >= 2:
                    cnt[train[r, 1]] -= 1
                    train[r, 1] = it
                    cnt[it] += 1
                    break
        del ucnt
    train = np.unique(train, axis=0)  # distinct pairs (re-pointing can collide; negligible)
    n_test = int(round(train_links * test_frac / (1 - test_frac)))
    test_pos = _sample_links(rs, user_w, test_items, item_w[test_items], P, Q, n_test, beta)
    ones = lambda a: np.concatenate([a, np.ones((a.shape[0], 1), dtype=np.int64)], 1)  # noqa: E731
    train = ones(train)
    rs.shuffle(train)
    test = ones(test_pos)
    test_seen = train[:test.shape[0]].copy()                # data/create_val.py:30

    if content_len > 0:
        # item text: tokens drawn from an item-factor-conditioned multinomial, zero LEFT-padded.
        # text_scale sets how peaked an item's word distribution is: 0.65 gives ~0.7 distinct words
        # per token, CiteULike-a's ratio (8,000-word vocabulary, ~66 distinct of ~93 tokens per
        # abstract); larger values repeat words more (1.5: ~0.16), which is cheaper for every kernel
        # that works per distinct word.
        W = rs.normal(0, 1.0, (vocab, rank))
        C = np.zeros((items, content_len), dtype=np.int64)
        lens = rs.randint(content_len // 4, content_len + 1, items)
        for i0 in range(0, items, 1000):
            i1 = min(items, i0 + 1000)
            logits = text_scale * (Q[i0:i1] @ W[1:].T)
            pr = np.exp(logits - logits.max(axis=1, keepdims=True))
            cdf = np.cumsum(pr, axis=1)
            cdf /= cdf[:, -1:]
            u = rs.random_sample((i1 - i0, content_len))
            for r, it in enumerate(range(i0, i1)):                 # inverse-CDF draws, one item at a time
                toks = np.minimum(np.searchsorted(cdf[r], u[r, :lens[it]], side="right"), vocab - 2) + 1
                C[it, content_len - lens[it]:] = toks
    else:
        C = np.zeros((items, 1), dtype=np.int64)
    return {"C": C, "train_items": train_items, "test_items": test_items, "train": train, "test": test,
            "test_seen": test_seen}


def data_spec_of(data):
    """DataSpec as get_data builds it (configs/data_utils.py:23-35)."""
    train, test, C = data["train"], data["test"], data["C"]
    user_count = int(max(np.max(train[:, 0]), np.max(test[:, 0])) + 1)
    return DataSpec(user_count, int(np.max(C) + 1), int(C.shape[0]), int(C.shape[1]))
