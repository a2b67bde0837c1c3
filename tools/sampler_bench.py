#!/usr/bin/env python3
"""Samples/s of the negative sampler (rows a1-a4): HIP sample kernel vs the compiled reference
(oracle/_ref, when present) / the C restatement on one host core.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nncf_amd.sampler import MultinomialSampler
    from nncf_amd import engine as nengine
    from oracle import sampler_oracle as so
    rs = np.random.RandomState(0)
    dist = (rs.zipf(1.3, 17000) % 5000 + 1).astype(np.float64)      # CiteULike-shaped item degrees
    s = MultinomialSampler(dist, dist.size, 1.0, 12345)
    out = {}
    for n in (5120, 1 << 20, 1 << 26):
        buf = torch.empty(n, dtype=torch.int32, device="cuda")
        s.sample_batch_device(n, out=buf)
        torch.cuda.synchronize()
        reps = 200 if n <= 5120 else 20
        nengine.profile_begin("sample_kernel", reps)
        t0 = time.perf_counter()
        for _ in range(reps):
            s.sample_batch_device(n, out=buf)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        raw_ms, launches, ov_ms = nengine.profile_end()
        kern = max(raw_ms - ov_ms, 1e-9) * 1e-3
        out["n=%d" % n] = {"kernel_us": kern * 1e6, "wall_us": wall * 1e6, "samples_per_s_kernel": n / kern,
                           "write_GBps": 4.0 * n / kern / 1e9, "frac_of_8TBps": 4.0 * n / kern / 8e12}
    kind = "reference" if so.ref_available() else "port"
    cpu = (so.RefSampler if kind == "reference" else so.OracleSampler)(dist, 1.0, 12345)
    n = 1 << 22
    cpu.sample_batch(1024)
    t0 = time.perf_counter()
    cpu.sample_batch(n)
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"kind": kind, "cores": 1, "samples_per_s": n / dt,
                           "sample": "%d draws, table build excluded" % n}
    print(json.dumps({"metric": "negative samples/s (unigram table, 17K ids)", **out}))


if __name__ == "__main__":
    main()
