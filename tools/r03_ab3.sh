#!/bin/bash
# third Winograd form, experiment variants (SISIC_WINO_VAR: bit 0 B operands one pair ahead, bit 1 halo three chunks ahead)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for rep in 1 2; do
for var in 0 1 2 3; do
  echo "=== rep $rep SISIC_WINO_VAR=$var"
  SISIC_WINO_VAR=$var timeout -k 10 300 python tools/conv_bench.py --cfgs 71 --match "@64 gn" --iters 40 2>&1 | grep " 71 " || exit 1
  SISIC_WINO_VAR=$var timeout -k 10 300 python tools/conv_bench.py --cfgs 70 --match "@32 gn" --iters 40 2>&1 | grep " 70 " || exit 1
  SISIC_WINO_VAR=$var timeout -k 10 300 python tools/conv_bench.py --cfgs 70 --match "@16 gn" --iters 40 2>&1 | grep " 70 " || exit 1
done
done > gpurun_out/ab_var.txt
cat gpurun_out/ab_var.txt
