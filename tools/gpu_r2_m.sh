#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -q -x -k "winograd or partials or fuzz or golden or latency or forward" > gpurun_out/pytest_part.log 2>&1; echo "part rc=$?"; tail -4 gpurun_out/pytest_part.log
if grep -q "Memory access fault" gpurun_out/pytest_part.log; then exit 1; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_s200_epi.json 2> gpurun_out/bench_s200_epi.log || echo "bench failed"
tail -2 gpurun_out/bench_s200_epi.log
timeout -k 10 300 python tools/conv_bench.py --cfgs 66,68,69 --match " gn" 2>&1 | grep -v "amdgpu.ids" | grep -E " 68 | 69 " | head -40
