#!/bin/bash
# bf16x3 pointwise kernel (tile_cfg 28; 29 / 30 force its 32- / 64-pixel form): parity, the two forms bit-equal, then the
# UNet's 1x1 layers against the f32 pointwise kernel (tile_cfg 20)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "1x1 or pointwise" > gpurun_out/ab9_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/ab9_pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/ab9_pytest.log | head -20; exit 1; fi
timeout -k 10 300 python tools/pwb_forms_check.py 2>&1 | grep -v amdgpu | tail -3 || exit 1
timeout -k 10 300 python tools/conv_bench.py --cfgs 20,28,29,30 --match "1x1" --iters 30 2>&1 | grep -v "amdgpu\|best cfg" > gpurun_out/ab_pwb.txt || exit 1
cat gpurun_out/ab_pwb.txt
