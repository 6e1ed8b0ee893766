#!/bin/bash
# in-place K-split reduction (tile_cfg 78 / 79 / 91): parity, per-layer time, whole step, single-image latency
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ksplit or winograd or partials" > gpurun_out/pytest_inplace.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_inplace.log
if grep -q "Memory access fault" gpurun_out/pytest_inplace.log; then exit 1; fi
grep -q "failed" gpurun_out/pytest_inplace.log && exit 1
timeout -k 10 300 python tools/conv_bench.py --batch 64 --match "@8" --cfgs 90,92,91,0 > gpurun_out/conv_bench_91.txt 2>&1 || exit 1
grep -v "amdgpu.ids" gpurun_out/conv_bench_91.txt
timeout -k 10 300 python tools/conv_bench.py --batch 1 --scale 2 --match "gn+res" --cfgs 76,77,78,79 > gpurun_out/conv_bench_78.txt 2>&1 || exit 1
grep -v "amdgpu.ids" gpurun_out/conv_bench_78.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_inplace.json 2> gpurun_out/bench_inplace.log || exit 1
tail -2 gpurun_out/bench_inplace.log
timeout -k 10 300 python tools/latency_one.py 1 128 1 > gpurun_out/latency_inplace.txt 2>&1 || exit 1
timeout -k 10 300 python tools/latency_one.py 1 64 1 >> gpurun_out/latency_inplace.txt 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/latency_inplace.txt
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sampler.py tests/test_gpu_classifier.py tests/test_gpu_train.py -m gpu -q -x > gpurun_out/pytest_inplace_b.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_inplace_b.log
