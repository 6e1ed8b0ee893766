#!/bin/bash
# start-delay stagger of the two workgroups of a CU (third Winograd form, 64-channel geometry): per-layer timings
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for pct in 0 50 100 150 200; do
  echo "=== SISIC_WINO_DELAY=$pct"
  SISIC_WINO_DELAY=$pct timeout -k 10 300 python tools/conv_bench.py --cfgs 71 --match "@64" --iters 30 2>&1 | grep -v amdgpu.ids | grep " 71 " || exit 1
done > gpurun_out/ab_delay.txt
cat gpurun_out/ab_delay.txt
