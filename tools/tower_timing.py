"""Developer tool: time basic_embedding neg_shared steps at CiteULike shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nncf_amd.conf import get_conf
from nncf_amd.data_utils import get_data
from nncf_amd.model_framework import get_model
I, U, L, W = 17000, 16000, 200, 8001
rs = np.random.RandomState(0)
C = rs.randint(1, W, (I, L)); 
for i in range(I): C[i, :rs.randint(0, 150)] = 0
n = 200000
train = np.stack([rs.randint(0, U, n), rs.randint(0, I, n), np.ones(n, dtype=np.int64)], 1)
data = {"C": C, "train": train, "test": train[:1000].copy(), "test_seen": train[:1000].copy(), "train_items": np.arange(I), "test_items": np.arange(I)}
pd = {"reset_after_getconf": True, "max_epoch": 1, "loss": "skip-gram", "num_negatives": 0, "neg_loss_weight": 128, "loss_gamma": 10,
      "learn_rate": 0.01, "batch_size_p": 512, "user_dim": 100, "item_dim": 100, "word_dim": 100, "word_emb_dropout_rate": 0.}
conf = get_conf("synthetic", "best", dict({"optimizer": "sgd"}, **pd), model_choice="basic_embedding")
helper = get_data(data, conf)
md = get_model(conf, helper, "basic_embedding")
head = md["model_neg_shared"]
e = head.engine
uid, cid = e.ids(train[:, 0]), e.ids(train[:, 1])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
head.run_epoch(uid, cid, 20); torch.cuda.synchronize()
t0 = time.perf_counter(); l = head.run_epoch(uid[20*512:], cid[20*512:], steps); torch.cuda.synchronize(); t1 = time.perf_counter()
print("basic_embedding neg_shared: %.1f us/step, mean loss %.4f" % ((t1 - t0) / steps * 1e6, l.mean().item()))
