#!/bin/bash
# tile_cfg 91 (8x8 level on the second geometry, two images per workgroup): parity, per-layer time, whole step
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ksplit or winograd" > gpurun_out/pytest_91.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_91.log
if grep -q "Memory access fault" gpurun_out/pytest_91.log; then exit 1; fi
grep -q " passed" gpurun_out/pytest_91.log || exit 1
grep -q "failed" gpurun_out/pytest_91.log && exit 1
timeout -k 10 300 python tools/conv_bench.py --batch 64 --match "@8" --cfgs 90,91,0 > gpurun_out/conv_bench_91.txt 2>&1 || exit 1
grep -v "amdgpu.ids" gpurun_out/conv_bench_91.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_91.json 2> gpurun_out/bench_91.log || exit 1
tail -2 gpurun_out/bench_91.log
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sampler.py tests/test_gpu_classifier.py -m gpu -q -x > gpurun_out/pytest_91b.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_91b.log
