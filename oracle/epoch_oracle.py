"""TEST INFRASTRUCTURE (checker only: imported by tests/, tools/convergence_parity.py and bench.py's reporting legs; never by
nncf_amd/).  The reference's trainer loops written out over the step oracle, epoch by epoch, and the whole@k evaluator in
user chunks -- what the product `Trainer` is compared with for WHOLE epochs at the BASELINE shape.

Reference loops restated here:
  models/train_neg_shared.py:34-78        shuffle -> consecutive batches of batch_size_p links -> train_on_batch, ragged tail dropped
  models/train_group_neg_shared.py:42-68  group_shuffle_train(by item, chop) -> the same
  models/train_original.py:34-88          shuffle -> batch + k negatives per positive from the sampler -> train_on_batch
  utils/objectives.py:395-421,296-370     whole@k: candidates = sorted(set(items)) of the split, per-user top-k, mean over users with truth
  utils/objectives.py:470-491             Evaluator.run: train split then test split
"""
import numpy as np

from . import step_oracle as so


def epoch_batches(train, scheme, b, shuffle, group_shuffle=None):
    """One epoch's link order as the reference trainers produce it (consumes np.random exactly as they do) and the batches
    that are trained: whole batches only (train_neg_shared.py:46-48: `if len(batch) < batch_size_p: break`)."""
    if scheme == "group_neg_shared":
        train = group_shuffle(train)
    else:
        shuffle(train)
    n_steps = len(train) // b
    return train, [train[t * b:(t + 1) * b] for t in range(n_steps)]


def run_reference_epoch(orc, scheme, batches, step_scale=None, sampler=None, k=0, loss_name="skip-gram", on_step=None):
    """Train `orc` (a StepOracle) over `batches` like the reference's inner loop.  step_scale(t) -> per-word scale of the word
    Embedding's row dropout at the t-th step of this epoch (None: no dropout) -- the product's masks are a pure function of
    (seed, train-forward index), so the checker asks the product for them and feeds the oracle the same masks.
    Returns the per-step losses."""
    out = np.zeros(len(batches))
    for t, bt in enumerate(batches):
        if step_scale is not None:
            orc.row_scale = step_scale(t)
        if scheme == "original":
            u, c, y = so.assemble_original(bt, sampler.sample_batch(k * len(bt)), loss_name)
            out[t] = orc.step_original(u, c, y)
        elif scheme == "neg_shared":
            out[t] = orc.step_neg_shared(bt[:, 0], bt[:, 1])
        elif scheme == "group_neg_shared":
            out[t] = orc.step_group_neg_shared(bt[:, 0], bt[:, 1])
        else:
            raise ValueError(scheme)
        if on_step is not None:
            on_step(t, out[t])
    return out


def inference_tables(orc, n_users, n_items, chunk=2048):
    """(U_hat, V_hat) in float64 as predict_on_batch sees them (objectives.py:296-321: Keras inference phase -- BatchNorm
    running statistics, no dropout; l2_normalize as configured, model_framework.py:62-63,109-111)."""
    o64 = so.StepOracle(orc.U, orc.V if orc.tower is None else None, orc.conf, dtype=np.float64)
    _, uh, _, _ = o64._user_side(np.arange(n_users))
    if orc.tower is None:
        _, vh, _ = o64._item_side(np.arange(n_items))
        return uh, vh
    rows = [np.asarray(orc.tower.forward(np.arange(i0, min(n_items, i0 + chunk)), train=False), dtype=np.float64)
            for i0 in range(0, n_items, chunk)]
    o64.V = np.concatenate(rows, 0)
    _, vh, _ = o64._item_side(np.arange(n_items))
    return uh, vh


def evaluate_whole_chunked(uh, vh, links, split_links, topk, user_chunk=1024):
    """so.evaluate_whole (objectives.py:395-421,296-370; metrics_ranking.py:6-35) without the dense (users x candidates) truth
    matrix: candidates = sorted(set(split_links[:, 1])); truth from `links` (the positives of the split) whose item is a
    candidate; users without truth are skipped; mean (map, recall, precision) over the rest, and per-user values."""
    nu, ni = uh.shape[0], vh.shape[0]
    cand = np.unique(split_links[:, 1])
    slot = -np.ones(ni, dtype=np.int64)
    slot[cand] = np.arange(len(cand))
    lk = links[slot[links[:, 1]] >= 0]
    lk = lk[np.argsort(lk[:, 0], kind="stable")]
    starts = np.searchsorted(lk[:, 0], np.arange(nu + 1))
    vc = vh[cand]
    per_user = np.full((nu, 3), np.nan)
    for u0 in range(0, nu, user_chunk):
        pred = uh[u0:u0 + user_chunk] @ vc.T
        for u in range(u0, min(nu, u0 + user_chunk)):
            if starts[u + 1] == starts[u]:
                continue
            truth = np.zeros(len(cand))
            truth[slot[lk[starts[u]:starts[u + 1], 1]]] = 1
            per_user[u] = so.eval_multiple(truth, pred[u - u0], topk)
    ok = ~np.isnan(per_user[:, 0])
    m = per_user[ok].mean(axis=0)
    return dict(map=float(m[0]), recall=float(m[1]), precision=float(m[2]), users=int(ok.sum())), per_user


def evaluate_run(orc, data, topk, n_users, n_items):
    """Evaluator.run(eval_scheme='whole') on the oracle's current state: (train results, test results)."""
    uh, vh = inference_tables(orc, n_users, n_items)
    tr, te = data["train"], data["test"]
    r_tr, _ = evaluate_whole_chunked(uh, vh, tr, tr, topk)
    r_te, _ = evaluate_whole_chunked(uh, vh, te[te[:, 2] == 1], te, topk)
    return r_tr, r_te
