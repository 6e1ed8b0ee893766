#!/bin/bash
# Which part of a kernel costs what: diagnostic builds of conv_winograd.hip with parts of a loop compiled out (a -D<MACRO>=<bits>
# per build: e.g. BF3_VAR (bits) or BF3_ONE_UA (0 / 1) for conv_winograd_bf3.inc; the values are listed beside the macro's
# #ifndef), timed per layer with tools/conv_bench.py.  Results of these builds are WRONG: timing only, never shipped.
#   build (here, no GPU):  bash tools/kernel_variants.sh build BF3_VAR "0 1 2 4 8 16 32"
#   run (GPU box):         bash tools/kernel_variants.sh run BF3_VAR "0 1 2 4 8 16 32" 74 "64->64 @64 gn+res"   -> gpurun_out/variants_BF3_VAR.txt
# (replaces round 3's one-off tools/r03_ab*.sh / r03_bf3_var.sh)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mode=$1; macro=$2; vars=$3; cfg=${4:-75}; match=${5:-"64 @64 gn"}
if [ "$mode" = build ]; then
    mkdir -p tools/bin
    for v in $vars; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$macro=$v -c synt_isic_amd/csrc/conv_winograd.hip -o /tmp/var_${macro}_$v.o || exit 1
        objs=$(ls synt_isic_amd/csrc/build/*.o | grep -v conv_winograd.o)
        /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/var_${macro}_$v.o -o tools/bin/libsisic_${macro}_$v.so || exit 1
    done
    exit 0
fi
mkdir -p gpurun_out
out=gpurun_out/variants_$macro.txt
: > $out
for v in $vars; do
    echo "=== $macro=$v  (tile_cfg $cfg, layers matching '$match')" >> $out
    SISIC_LIB_PATH=$PWD/tools/bin/libsisic_${macro}_$v.so timeout -k 10 200 python tools/conv_bench.py --cfgs $cfg --match "$match" --iters 30 2>&1 | grep " $cfg " >> $out || exit 1
done
cat $out
