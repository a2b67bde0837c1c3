#!/bin/bash
# Developer loop: device-only compile of ONE .hip file with the compiler's resource remarks, one line per kernel:
#   bash tools/kres.sh nncf_amd/csrc/cnn.hip [name filter]
# (the committed audit of the shipped library is tools/isa_resources.py)
F=$1; PAT=${2:-.}
D=$(dirname $F)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 --cuda-device-only -I$D/../../include -c $F -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"sym": m.group(1)}; rows.append(cur); continue
    if "error" in line and "remark" not in line: print(line, end="")
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1).split(" [")[0]] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["sym"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%5s %5s %5s %6s %6s %7s %4s %7s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "occ", "lds", "kernel"))
pat = sys.argv[1]
for r, n in zip(rows, names):
    n = re.sub(r"\(.*\)$", "", n.replace("void ", ""))
    if re.search(pat, n):
        print("%5d %5d %5d %6d %6d %7d %4d %7d  %s" % (r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("LDS Size", 0), n))
' "$PAT"
rm -f /tmp/kres_$$.o
