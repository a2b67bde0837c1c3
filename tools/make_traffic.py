#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries tools/profile_round.sh leaves behind.

    python tools/make_traffic.py <dir with <tag>_*_pmc_fetch.txt / _pmc_write.txt> <tag> <out.json>

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are KiB per dispatch;
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it counts 32-B requests as
half of what 16-B/lane coalesced reads move)."""
import json
import os
import sys

# bench key -> (profile name stem, kernel whose traffic bench.py reports)
CASES = {
    "basic_embedding/neg_shared/skip-gram/b512/d100": ("basic_embedding_neg_shared", "fused_strip_kernel"),
    "mf/neg_shared/skip-gram/b512/d100": ("mf_neg_shared", "fused_strip_kernel"),
}


def pmc_mean(path, kernel, counter):
    if not os.path.exists(path):
        return None
    seen = False
    for line in open(path):
        if line.startswith("# PMC counters"):
            seen = True
            continue
        if seen and kernel in line and counter in line:
            return float(line.split()[-1])
    return None


def trace_avg_us(path, kernel):
    """average duration (us) of `kernel` (its most-called template variant) in a tools/prof_summary.py kernel-trace summary"""
    ent = all_kernel_avgs(path).get(kernel)
    return ent["rocprof_avg_us"] if ent else None


def all_kernel_avgs(path):
    """{kernel: avg_us} of every kernel in a kernel-trace summary; of a kernel's template variants the one with the MOST calls
    (the 4-call inference variant of tower_pool_kernel used to overwrite the 5,121-call training one)"""
    res, calls_of = {}, {}
    if not os.path.exists(path):
        return res
    for line in open(path):
        if line.startswith("# PMC counters"):
            break
        parts = line.split()
        tok = [x for x in parts[:2] if "_kernel" in x]     # "nncf::x_kernel ..." or "void nncf::x_kernel<...> ..."
        if len(parts) >= 5 and tok:
            try:
                name = tok[0].split("<")[0].split("::")[-1]
                calls, avg = int(parts[-4]), float(parts[-2])
            except ValueError:
                continue
            if calls > calls_of.get(name, -1):
                calls_of[name] = calls
                res[name] = {"rocprof_avg_us": avg, "calls": calls, "variant": line[:72].strip()}
    return res


def source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    return g._source_hash()


def evaluator_entry(d, tag):
    """whole@50 pass of tools/eval_bench.py (train + test candidate sets = two launches of eval_fused_kernel): the PMC means are
    per dispatch over both launch kinds, a pass is two dispatches."""
    f = pmc_mean(os.path.join(d, "%s_eval_pmc_fetch.txt" % tag), "eval_fused_kernel", "FETCH_SIZE")
    w = pmc_mean(os.path.join(d, "%s_eval_pmc_write.txt" % tag), "eval_fused_kernel", "WRITE_SIZE")
    if f is None or w is None:
        return None
    per_launch = int(round((2 * f + w) * 1024))
    return {"kernel": "eval_fused_kernel", "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "bytes_per_launch": per_launch,
            "launches_per_pass": 2, "bytes_per_pass": 2 * per_launch,
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over tools/eval_bench.py, separate passes "
                   "(profiles/%s_eval_pmc_*.txt); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch, mean over the train- and "
                   "the test-candidate launches; a whole@50 pass is one of each" % tag,
            "round": int(tag.lstrip("r") or 0), "csrc_hash": source_hash()}


def main():
    d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
    table = {}
    ev = evaluator_entry(d, tag)
    if ev:
        table["evaluator/whole@50/16Kx17K/d100"] = ev
    for key, (stem, kernel) in CASES.items():
        f = pmc_mean(os.path.join(d, "%s_%s_pmc_fetch.txt" % (tag, stem)), kernel, "FETCH_SIZE")
        w = pmc_mean(os.path.join(d, "%s_%s_pmc_write.txt" % (tag, stem)), kernel, "WRITE_SIZE")
        if f is None or w is None:
            continue
        table[key] = {
            "kernel": kernel, "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
            "bytes_per_launch": int(round((2 * f + w) * 1024)),
            "rocprof_avg_us": trace_avg_us(os.path.join(d, "%s_%s_kernel_trace.txt" % (tag, stem)), kernel),
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes "
                   "(profiles/%s_%s_pmc_*.txt); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024" % (tag, stem),
            "round": int(tag.lstrip("r") or 0),
            "kernels": all_kernel_avgs(os.path.join(d, "%s_%s_kernel_trace.txt" % (tag, stem))),
            # bench.py trusts these durations only while the kernel sources are the ones that were profiled
            "csrc_hash": source_hash(),
        }
    json.dump(table, open(out, "w"), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
