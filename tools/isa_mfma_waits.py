#!/usr/bin/env python3
"""Which kernels of nncf_amd/libnncf_hip.so wait for an LDS / memory read in the middle of an MFMA sequence?  For every kernel of
the shipped gfx950 code objects: the longest runs of v_mfma instructions and, between the first and the last MFMA of every
basic-block-free stretch, the blocking waits (s_waitcnt lgkmcnt(0) / vmcnt(0)) the compiler put between them.  A 'read two k-steps,
wait, four MFMAs' pattern (csrc/eval.hip before round 6) shows as one blocking wait per few MFMAs."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_resources as ir  # noqa: E402


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ir.ROOT, "nncf_amd", "libnncf_hip.so")
    rows = []
    for co in ir.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            path = f.name
        dis = subprocess.run([os.path.join(ir.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
        os.unlink(path)
        cur, ins = None, {}
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                cur = m.group(1)
                ins[cur] = []
            elif cur is not None and line.strip():
                ins[cur].append(line.strip().split("//")[0].strip())
        for name, code in ins.items():
            mf = [i for i, s in enumerate(code) if s.startswith("v_mfma")]
            if not mf:
                continue
            # stretches of MFMAs with no branch between them
            stretches, start = [], mf[0]
            prev = mf[0]
            for i in mf[1:]:
                if any(c.startswith("s_cbranch") or c.startswith("s_branch") or c.startswith("s_barrier") for c in code[prev:i]):
                    stretches.append((start, prev))
                    start = i
                prev = i
            stretches.append((start, prev))
            worst = None
            for a, b in stretches:
                n = sum(1 for c in code[a:b + 1] if c.startswith("v_mfma"))
                if n < 8:
                    continue
                blk = sum(1 for c in code[a:b + 1] if c.startswith("s_waitcnt") and ("lgkmcnt(0)" in c or "vmcnt(0)" in c))
                if worst is None or blk / n > worst[1] / worst[0]:
                    worst = (n, blk)
            if worst:
                rows.append((worst[1] / worst[0], worst[0], worst[1], len(mf), name))
    names = ir._demangle([r[4] for r in rows])
    print("blocking waits inside the MFMA stretch with the most of them per MFMA (stretches of >= 8 MFMAs without a branch / barrier)")
    print("%8s %6s %6s %6s  %s" % ("per mfma", "mfma", "waits", "total", "kernel"))
    for (r, n, b, t, _), nm in sorted(zip(rows, names), key=lambda x: -x[0][0]):
        print("%8.2f %6d %6d %6d  %s" % (r, n, b, t, nm[:150]))


if __name__ == "__main__":
    main()
