#!/usr/bin/env python3
"""Generate tests/golden/sampler_*.npz from the COMPILED, UNMODIFIED reference sampler.

Runs only in the build container (needs /root/reference to build oracle/_ref).  Nothing
of the reference's source is written out -- only inputs (degree vectors, powers, LCG
state0) and the outputs the reference produced for them (boundary arrays, sample
streams, a SHA-256 of a long stream).

Usage:  python tools/gen_golden_sampler.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sampler_oracle as so  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def zipf_degrees(v, seed, alpha=1.0, total=200000):
    """Integer Zipf(alpha)-shaped degree histogram over a random id permutation; >=1 each."""
    rs = np.random.RandomState(seed)
    w = 1.0 / np.arange(1, v + 1) ** alpha
    deg = np.maximum(1, np.floor(w / w.sum() * total)).astype(np.float64)
    return deg[rs.permutation(v)]


def cases():
    c = {}
    c["toy"] = np.array([5., 3., 0., 2.])
    v = 17000
    i = np.arange(v)
    c["mod13"] = np.where(i % 7 == 0, 0., 1. + (i % 13)).astype(np.float64)
    c["zipf17000"] = zipf_degrees(17000, 1)
    c["zipf16000"] = zipf_degrees(16000, 2)
    c["uniform1000"] = np.ones(1000)
    c["single"] = np.array([0., 0., 4., 0.])
    lead = zipf_degrees(5000, 3)
    lead[:100] = 0.  # leading zero-degree block
    lead[-50:] = 0.  # trailing zero-degree block
    c["zeros_edges5000"] = lead
    return c


def main():
    so.build()
    assert so.ref_available(), "needs oracle/_ref (build container with /root/reference)"
    os.makedirs(GOLD, exist_ok=True)
    states = [1, 42, 12345, 2**63 + 5]
    powers = [1.0, 0.75]
    out = {}
    meta = []
    # LCG known answers (nodesampler.cpp:23-26) via the reference object itself: a
    # uniform dist over 1e8 ids would be the identity table, too big; instead use the
    # identity-on-buckets trick: dist of size 4 with equal degree => id = r // 25e6.
    # The raw r stream is cross-checked against big-int Python arithmetic separately.
    for name, dist in cases().items():
        for p in powers:
            ref = so.RefSampler(dist, power=p, state0=1)
            tag = "%s_p%s" % (name, ("%g" % p).replace(".", "_"))
            out[tag + "_dist"] = dist
            out[tag + "_bounds"] = ref.bounds()
            for s0 in states:
                ref.state = s0
                out["%s_s%d_first" % (tag, s0)] = ref.sample_batch(4096)
                out["%s_s%d_state_after" % (tag, s0)] = np.array([ref.state], dtype=np.uint64)
                # single-draw API continues the same stream (nodesampler.cpp:69-71)
                out["%s_s%d_single" % (tag, s0)] = np.array([ref.sample() for _ in range(16)], dtype=np.int32)
            if name in ("zipf17000", "mod13") and p == 1.0:
                ref.state = 12345
                h = hashlib.sha256()
                for _ in range(10):
                    h.update(ref.sample_batch(1000000).tobytes())
                out[tag + "_s12345_sha256_1e7"] = np.frombuffer(h.digest(), dtype=np.uint8)
            meta.append(tag)
            ref.close()
    out["tags"] = np.array(meta)
    out["states"] = np.array(states, dtype=np.uint64)
    np.savez_compressed(os.path.join(GOLD, "sampler_golden.npz"), **out)
    sz = os.path.getsize(os.path.join(GOLD, "sampler_golden.npz"))
    print("wrote sampler_golden.npz  %d arrays  %.1f KB" % (len(out), sz / 1024.))


if __name__ == "__main__":
    main()
