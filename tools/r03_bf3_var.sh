#!/bin/bash
# Which part of the bf16x3 Winograd kernel costs what: diagnostic builds with parts of the loop removed (BF3_VAR bits:
# 1 no A loads, 2 no transform, 4 no staging, 8 no MFMA, 16 no output), timed per layer.  Results of these builds are wrong.
#   build (here):  bash tools/r03_bf3_var.sh build     run (GPU box): bash tools/r03_bf3_var.sh
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
VARS="${BF3_VARS:-0 1 2 4 8 16 31 23 27 29}"
if [ "$1" = build ]; then
    mkdir -p tools/bin
    for v in $VARS; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBF3_VAR=$v -c synt_isic_amd/csrc/conv_winograd.hip -o /tmp/bf3_var_$v.o || exit 1
        objs=$(ls synt_isic_amd/csrc/build/*.o | grep -v conv_winograd.o)
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/bf3_var_$v.o -o tools/bin/libsisic_bf3_var$v.so || exit 1
    done
    exit 0
fi
mkdir -p gpurun_out
: > gpurun_out/bf3_var.txt
for v in $VARS; do
    echo "=== BF3_VAR=$v" >> gpurun_out/bf3_var.txt
    SISIC_LIB_PATH=$PWD/tools/bin/libsisic_bf3_var$v.so timeout -k 10 200 python tools/conv_bench.py --cfgs 74 --match "${BF3_MATCH:-64 @64 gn}" --iters 30 2>&1 | grep " 74 " >> gpurun_out/bf3_var.txt || exit 1
done
cat gpurun_out/bf3_var.txt
