"""Developer tool: per-phase wall-clock stamps (10 ns ticks) of the conv forward / backward workgroups of one cnn_embedding
group_neg_shared step at the CiteULike shape.  NNCF_CNN_DBG=64 python tools/cnn_timing.py"""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("NNCF_CNN_DBG", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nncf_amd import _lib, synthetic
from nncf_amd.conf import get_conf
from nncf_amd.data_utils import get_data
from nncf_amd.model_framework import get_model

data = synthetic.make_citeulike_like(content_len=200, vocab=8001)
pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "log-loss", "num_negatives": 0, "neg_loss_weight": 128, "loss_gamma": 10,
      "learn_rate": 0.01, "batch_size_p": 512, "user_dim": 100, "item_dim": 100, "word_dim": 100, "chop_size": 5,
      "train_scheme": "group_neg_shared", "word_emb_dropout_rate": 0.5, "num_filters": [100], "filter_lengths": [3],
      "conv_dropout_rate": 0.1}
conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="cnn_embedding")
helper = get_data(dict(data), conf, rand_seed=5)
md = get_model(conf, helper, "cnn_embedding", seed=2)
head = md["model_group_neg_shared"]
tw = md["model"].core.tower
tr = data["train"]
rs = np.random.RandomState(0)
items = rs.choice(tr[:, 1].max() + 1, 125, replace=False)
cid = np.repeat(items, 5)[:512]
cid = np.concatenate([cid, items[:512 - cid.size]])
uid = rs.randint(0, tr[:, 0].max() + 1, 512)
for _ in range(5):
    head.train_on_batch([uid, cid], [np.ones(512)])
torch.cuda.synchronize()
ngrp = (((tw.F + 15) // 16) + 1) // 2
print('F', tw.F, 'fl', tw.filter_lens, 'groups', ngrp, 'cap', tw.cap)
ucap = tw.cap + min(tw.cap, 512) // 2   # slots + split halves (cnn_units_build)
n = 2 * ucap * ngrp * 16
out = np.zeros(n, dtype=np.int64)
_lib.check(tw._L.nncf_tower_debug_stamps(tw.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), out.size))
st = out.reshape(2, ucap * ngrp, 16)
nwg = ucap * ngrp
names = [["start", "index", "rows->LDS", "P gemm", "assembly", "Y stores", "stats"],
         ["start", "index", "dY tile", "rows + S", "(b) dWk", "(a) dW rows", "stores"]]
for k, kn in enumerate(("cnn_conv_fwd", "cnn_conv_bwd")):
    raw = st[k, :nwg, :7]
    raw = raw[raw[:, 0] != 0]
    s = (raw - raw[:, 0].min()).astype(np.float64) * 0.01   # us
    nwg = len(s)
    t0 = s[:, 0].min()
    print("%s: %d workgroups; first start 0, last start %.2f us, last end %.2f us" % (kn, nwg, s[:, 0].max() - t0, s[:, 6].max() - t0))
    for i in range(1, 7):
        d = s[:, i] - s[:, i - 1]
        print("  %-12s mean %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us   (reached at mean %6.2f us)" % (
            names[k][i], d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max(), (s[:, i] - t0).mean()))
