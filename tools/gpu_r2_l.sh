#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SISIC_LIB_PATH=$PWD/tools/bin/libsisic_hip_timing.so timeout -k 10 300 python tools/wino_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wino_phases.txt
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_unet.py -q -x > gpurun_out/pytest_train.log 2>&1; echo "train rc=$?"; tail -4 gpurun_out/pytest_train.log
if grep -q "Memory access fault" gpurun_out/pytest_train.log; then exit 1; fi
timeout -k 10 300 python tools/train_bench.py --steps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/train_bench.txt
