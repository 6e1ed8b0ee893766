#!/bin/bash
# training after the weight-gradient changes: tests, step time, kernel statistics
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_classifier.py -m gpu -q -x > gpurun_out/pytest_train.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_train.log
if grep -q "Memory access fault" gpurun_out/pytest_train.log; then exit 1; fi
grep -q "failed\|error" gpurun_out/pytest_train.log && exit 1
timeout -k 10 300 python tools/train_bench.py --steps 5 > gpurun_out/train_bench.txt 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/train_bench.txt
bash tools/prof_script.sh trainp tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -30
