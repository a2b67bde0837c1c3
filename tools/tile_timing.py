"""Developer tool: per-phase cycle breakdown of shared_tile_kernel from in-kernel stamps.
NNCF_DEBUG_TIMING=1 python tools/tile_timing.py [loss]"""
import ctypes, os, sys
import numpy as np
os.environ["NNCF_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nncf_amd.engine import Engine
from nncf_amd import _lib
loss = sys.argv[1] if len(sys.argv) > 1 else "skip-gram"
b, d = 512, 100
eng = Engine(16000, 17000, d, b, 0, loss, 10.0, 128.0, 1e-6, 0.01, loss in ("log-loss", "max-margin"), False)
rs = np.random.RandomState(0)
uid = eng.ids(rs.randint(0, 16000, b * 50)); cid = eng.ids(rs.randint(0, 17000, b * 50))
eng.epoch_neg_shared(uid, cid, 50)
torch.cuda.synchronize()
nb = b // 32
out = np.zeros(nb * nb * 8, dtype=np.int64)
_lib.check(eng._L.nncf_model_debug_stamps(eng._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), out.size))
st = out.reshape(nb * nb, 8)[:, :7]
names = ["ids", "gather rows->LDS+sync", "(diag)+MFMA S", "epilogue+sync", "colsum", "back-GEMM+store"]
dd = np.diff(st, axis=1)
print("loss", loss, " per-WG phase cycles: mean / min / max")
for i, n in enumerate(names):
    print("  %-18s %8.0f %8d %8d" % (n, dd[:, i].mean(), dd[:, i].min(), dd[:, i].max()))
print("  per-WG total      %8.0f" % (st[:, 6] - st[:, 0]).mean())
print("  kernel span (first start -> last end): %d cycles; start skew %d" % (st[:, 6].max() - st[:, 0].min(), st[:, 0].max() - st[:, 0].min()))
