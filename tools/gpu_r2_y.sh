#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for r in 1 0; do
  echo "SISIC_WGRAD_ROWS32=$r"
  SISIC_WGRAD_ROWS32=$r timeout -k 10 300 python tools/train_bench.py --steps 5 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x > gpurun_out/pytest_train.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_train.log
bash tools/prof_script.sh trainp tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -8
