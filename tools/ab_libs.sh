#!/bin/bash
# A/B of two builds of the library on ONE box: tools/conv_bench.py per layer, the libraries alternated `reps` times.
#   bash tools/ab_libs.sh "<lib A> <lib B> ..." <cfgs> "<match>" [reps]      -> gpurun_out/ab_libs.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
libs=$1; cfgs=${2:-74}; match=${3:-"gn"}; reps=${4:-2}
mkdir -p gpurun_out
out=gpurun_out/ab_libs.txt
: > $out
for r in $(seq $reps); do
    for l in $libs; do
        echo "=== rep $r  $l" >> $out
        SISIC_LIB_PATH=$PWD/$l timeout -k 10 300 python tools/conv_bench.py --cfgs $cfgs --match "$match" --iters 30 >> $out 2>&1 || exit 1
    done
done
cat $out
