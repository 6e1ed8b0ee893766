#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x > gpurun_out/pytest_w.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/pytest_w.log
if grep -q "Memory access fault" gpurun_out/pytest_w.log; then exit 1; fi
timeout -k 10 300 python tools/train_bench.py --steps 5 2>&1 | grep -v amdgpu.ids | cut -c1-250
bash tools/prof_script.sh trainp tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -9
