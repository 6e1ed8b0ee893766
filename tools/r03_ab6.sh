#!/bin/bash
# bf16x3 Winograd form (tile_cfg 74): parity, then per-layer times against the third f32 form
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog_bf3.txt timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -k "winograd or partials" > gpurun_out/ab6_pytest.log 2>&1
rc=$?; tail -12 gpurun_out/ab6_pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/ab6_pytest.log | head -20; exit 1; fi
timeout -k 10 300 python tools/conv_bench.py --cfgs 71,74 --match "@64" --iters 30 2>&1 | grep -v amdgpu > gpurun_out/ab_bf3.txt || exit 1
timeout -k 10 300 python tools/conv_bench.py --cfgs 70,74 --match "@32 gn" --iters 30 2>&1 | grep -v amdgpu >> gpurun_out/ab_bf3.txt || exit 1
timeout -k 10 300 python tools/conv_bench.py --cfgs 70,74 --match "@16 gn" --iters 30 2>&1 | grep -v amdgpu >> gpurun_out/ab_bf3.txt || exit 1
grep -v "best cfg" gpurun_out/ab_bf3.txt
