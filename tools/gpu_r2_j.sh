#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/pytest_train.log 2>&1; echo "train rc=$?"; tail -4 gpurun_out/pytest_train.log
if grep -q "Memory access fault" gpurun_out/pytest_train.log; then exit 1; fi
timeout -k 10 300 python tools/train_bench.py --steps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/train_bench.txt
timeout -k 10 300 python tools/conv_bench.py --cfgs 0,22,24,25 --match probe1x1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv1x1_probe.txt
timeout -k 10 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/configs_4_5.jsonl | cut -c1-400
