"""Developer tool: per-phase breakdown of fused_strip_kernel from in-kernel cycle stamps.
python tools/fused_timing.py [loss] [b] [d]"""
import ctypes, os, sys
import numpy as np
os.environ["NNCF_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nncf_amd.engine import Engine
from nncf_amd import _lib
if len(sys.argv) > 2 and sys.argv[1] == "--load":      # stamps saved by bench.py --dump-stamps
    st = np.load(sys.argv[2])
    nblk = st.shape[0]
    loss = None
else:
    loss = sys.argv[1] if len(sys.argv) > 1 else "skip-gram"
if loss is not None:
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    eng = Engine(16000, 17000, d, b, 0, loss, 10.0, 128.0, 1e-6, 0.01, False, False)
    rs = np.random.RandomState(0)
    uid = eng.ids(rs.randint(0, 16000, b * 50)); cid = eng.ids(rs.randint(0, 17000, b * 50))
    eng.epoch_neg_shared(uid, cid, 50)
    torch.cuda.synchronize()
    mb = b // 16
    nblk = (2 * (mb + 8) * 8 + 1)
    out = np.zeros(nblk * 32, dtype=np.int64)
    _lib.check(eng._L.nncf_model_debug_stamps(eng._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), out.size))
    st = out.reshape(nblk, 32)
names = {0: "entry", 1: "ids/desc issued + barrier", 2: "stats", 3: "own rows stored", 4: "rows(0) issued"}
for s in range(4):
    names[5 + 4 * s] = "st%d rows->LDS+barrier" % s
    names[6 + 4 * s] = "st%d S mfma" % s
    names[7 + 4 * s] = "st%d elementwise+barrier" % s
    names[8 + 4 * s] = "st%d back mfma+barrier" % s
names[19] = "(wide) last back mfma issued"
names[20] = "out->LDS, loss partial"
names[23] = "slab store + release + ticket"
names[21] = "reducer: acquire + slab sum"
names[22] = "finalize"
names[24] = "wait for all readers"
names[25] = "stats: rows summed (registers)"
names[26] = "stats: LDS reduce + publish"
order = [0, 1, 25, 26, 2, 3, 4] + list(range(5, 21)) + [20, 23, 21, 24, 22]
order = [k for i, k in enumerate(order) if k not in order[:i]]
def report(title, live):
    if not live:
        return
    t0 = min(st[i, 0] for i in live)
    tend = max(st[i].max() for i in live)
    print("%s: live workgroups %d, first entry -> last stamp %d cycles; entry skew %d" % (
        title, len(live), tend - t0, max(st[i, 0] for i in live) - t0))
    for k in order[1:]:
        rows = [i for i in live if st[i, k] > 0]
        if not rows:
            continue
        vals = []
        for i in rows:
            cand = [kk for kk in order[:order.index(k)] if st[i, kk] > 0 and st[i, kk] <= st[i, k]]
            if not cand:
                continue
            pk = max(cand, key=lambda kk: st[i, kk])
            vals.append(st[i, k] - st[i, pk])
        vals = np.array(vals)
        print("  %-30s n %4d mean %7.1f  min %6d  max %6d" % (names[k], len(rows), vals.mean(), vals.min(), vals.max()))
    tot = np.array([st[i].max() - st[i, 0] for i in live])
    print("  per-WG total mean %.1f max %d" % (tot.mean(), tot.max()))


live = [i for i in range(nblk) if st[i, 0] > 0 and st[i, 20] > 0]
report("all", live)
if "--split" in sys.argv:      # --split P gu: strip kernel's block id -> (strip, part); U-pass strips first
    P, gu = int(sys.argv[sys.argv.index("--split") + 1]), int(sys.argv[sys.argv.index("--split") + 2])
    side = lambda i: ((i >> 3) // P) * 8 + (i & 7) < gu
    report("U-pass", [i for i in live if side(i)])
    report("V-pass", [i for i in live if not side(i)])
