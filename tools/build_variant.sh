#!/bin/bash
# A/B builds of the large-batch kernels: bash tools/build_variant.sh NAME "-DNNCF_WIDE_FENCES=0 ..." -> ab/libnncf_hip_NAME.so
# (only step_fused.hip / step_pair.hip are recompiled; the other objects are the in-tree ones).  On the GPU box:
#   cp ab/libnncf_hip_NAME.so nncf_amd/libnncf_hip.so
set -eu
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/variant_$NAME; mkdir -p $T $R/ab
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -I$R/include -I$R/nncf_amd/csrc"
for f in ${VARIANT_FILES:-step_fused step_pair}; do $CXX $FLAGS -c $R/nncf_amd/csrc/$f.hip -o $T/$f.o & done
wait
OBJS=""
for o in $R/nncf_amd/csrc/*.o; do b=$(basename $o); if [ -f $T/$b ]; then OBJS="$OBJS $T/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libnncf_hip_$NAME.so $OBJS
ls -la $R/ab/libnncf_hip_$NAME.so
