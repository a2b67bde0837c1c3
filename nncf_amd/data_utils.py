"""Host side of the sampling path: mirrors the parts of the reference's
configs/data_utils.py that sit on the hot path (get_data's sampler wiring :15-70,
get_sampler :193-215, group_shuffle_train :218-241) on top of the HIP sampler.

Data loading: the reference reads data/.../data_split_cold_item.pkl (:73-104), which is not
shipped (.MISSING_LARGE_BLOBS).  `get_data` accepts the same dict schema from a pickle path, or
`data_name` = 'synthetic_citeulike[...]' for the CiteULike-shaped generator
(nncf_amd/synthetic.py)."""
import pickle

import numpy as np

from . import synthetic


class DataHelper(object):
    """Container the reference uses only as an attribute bag (configs/data_helper.py is
    offline preprocessing; data_utils.py:74,103)."""

    def __init__(self):
        self.data = None
        self.data_spec = None
        self.sampler_dict = None


def degree_dist(ratings, column=1, neg_dist="unigram"):
    """Degree histogram the sampler is built from (data_utils.py:201-208): length max id + 1,
    'uniform' => every id that occurs gets weight 1; suffixes after '_' are ignored."""
    neg_dist = neg_dist.split("_")[0]
    assert neg_dist == "uniform" or neg_dist == "unigram", [neg_dist]
    col = np.asarray(ratings[:, column], dtype=np.int64)
    dist = np.bincount(col, minlength=int(col.max()) + 1).astype(np.float64)
    if neg_dist == "uniform":
        dist[dist > 0] = 1
    return np.ascontiguousarray(dist)


def get_sampler(ratings, neg_dist="unigram", neg_sampling_power=0.75, column=1, rand_seed=0, batch_mode=True,
                device=0):
    """Return a sampling function (data_utils.py:193-215): `sample_batch(n) -> int32[n]` or
    `sample() -> int`, drawing ids of `ratings[:, column]` proportional to degree**power.
    The returned bound method keeps its MultinomialSampler alive (`fn.__self__`)."""
    from .sampler import MultinomialSampler
    dist = degree_dist(ratings, column, neg_dist)
    s = MultinomialSampler(dist, dist.size, neg_sampling_power, rand_seed, device=device)
    return s.sample_batch if batch_mode else s.sample


class _LazySamplerDict(dict):
    """sampler_dict with the reference's four keys (data_utils.py:45-67).  The reference
    builds all four 400 MB tables up front although a trainer uses one; here each sampler
    is created on first access."""

    def __init__(self, factories):
        super(_LazySamplerDict, self).__init__()
        self._factories = factories

    def __getitem__(self, key):
        if not dict.__contains__(self, key):
            if key not in self._factories:
                raise KeyError(key)
            dict.__setitem__(self, key, self._factories[key]())
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return key in self._factories

    def keys(self):
        return self._factories.keys()


def _load(data_name, **synth_kw):
    if isinstance(data_name, dict):
        return data_name
    if data_name.startswith("synthetic"):
        return synthetic.make_citeulike_like(**synth_kw)
    with open(data_name, "rb") as fp:
        return pickle.load(fp, encoding="latin1")


def get_data(data_name, conf, reverse_samping=False, rand_seed=0, device=0, **synth_kw):
    """get_data (data_utils.py:15-70): data dict + DataSpec + sampler_dict
    {'sample_u','sample_batch_u' (user negatives, when reverse_samping), 'sample','sample_batch'}."""
    helper = DataHelper()
    helper.data = _load(data_name, **synth_kw)
    train = helper.data["train"]
    helper.data_spec = synthetic.data_spec_of(helper.data)
    if conf is not None:
        neg_dist = conf.neg_dist
        power = getattr(conf, "neg_sampling_power", 0.75)  # data_utils.py:40-44

        def mk(column, batch_mode):
            return lambda: get_sampler(train, neg_dist=neg_dist, neg_sampling_power=power, column=column,
                                       rand_seed=rand_seed, batch_mode=batch_mode, device=device)

        factories = {"sample": mk(1, False), "sample_batch": mk(1, True)}
        if reverse_samping:  # data_utils.py:47-57 (column-swapped train == sampling column 0)
            factories["sample_u"] = mk(0, False)
            factories["sample_batch_u"] = mk(0, True)
        helper.sampler_dict = _LazySamplerDict(factories)
    return helper


def group_shuffle_train(train, by="item", chop=0, iidx=None, rng=None):
    """Stratified shuffle (data_utils.py:218-241): links grouped by item (or user) in a random
    group order, then -- if chop > 0 -- cut into runs of `chop` rows that are shuffled as units.

    RNG consumption follows the reference call for call (shuffle(iidx) in place, shuffle(train)
    in place, shuffle of the chop blocks) on `rng` (default: the global np.random, as the
    reference).  One deliberate deviation: the reference's `argsort()` (:234) is an unstable
    quicksort whose within-group order depends on the NumPy build; a STABLE sort is used here
    so that the stream is reproducible.  Between-group order and group
    membership are identical."""
    rng = np.random if rng is None else rng
    col = 0 if by == "user" else 1
    if iidx is None:
        iidx = np.arange(np.max(train[:, col]) + 1)
    rng.shuffle(iidx)
    rng.shuffle(train)
    key = iidx[train[:, col]]
    train = train[np.argsort(key, kind="stable")]
    if chop > 0:
        bulk_len = (train.shape[0] // chop) * chop
        bulk = train[:bulk_len].reshape((-1, chop, train.shape[-1]))
        rng.shuffle(bulk)
        train = np.vstack([bulk.reshape((-1, train.shape[-1])), train[bulk_len:]])
    return train
