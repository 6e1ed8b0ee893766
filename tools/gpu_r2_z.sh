#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for w in 256 384 512 768 1024; do
  echo "SISIC_WGRAD_WGS=$w"
  SISIC_WGRAD_WGS=$w timeout -k 10 300 python tools/train_bench.py --steps 5 2>&1 | grep -v amdgpu.ids | cut -c1-230 || exit 1
done
