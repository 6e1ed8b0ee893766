#!/bin/bash
# batched weight re-layout after the optimizer step: training tests, step time with and without it
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x > gpurun_out/pytest_train.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_train.log
if grep -q "Memory access fault" gpurun_out/pytest_train.log; then exit 1; fi
grep -q "failed\|error" gpurun_out/pytest_train.log && exit 1
for r in 0 1; do
  SISIC_REPACK_BATCH=$r timeout -k 10 300 python tools/train_bench.py --steps 5 > gpurun_out/train_bench_repack$r.txt 2>&1 || exit 1
  echo "SISIC_REPACK_BATCH=$r"; grep -v amdgpu.ids gpurun_out/train_bench_repack$r.txt
done
