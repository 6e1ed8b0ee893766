#!/bin/bash
# image-pair K-split at the 8x8 level (tile_cfg 91) + latency-mode tile choices: full GPU suite, bench, single-image latency
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
if grep -q "Memory access fault" gpurun_out/pytest_gpu.log; then exit 1; fi
grep -q "failed" gpurun_out/pytest_gpu.log && exit 1
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_pair.json 2> gpurun_out/bench_pair.log || exit 1
tail -2 gpurun_out/bench_pair.log
timeout -k 10 300 python tools/latency_one.py 1 128 1 > gpurun_out/latency_pair.txt 2>&1 || exit 1
timeout -k 10 300 python tools/latency_one.py 1 64 1 >> gpurun_out/latency_pair.txt 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/latency_pair.txt
