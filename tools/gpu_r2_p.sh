#!/bin/bash
# latency-mode tile choices for the stride-2 and the conv_out convolutions; training-step kernel statistics of the current library
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "stride2 or small_cout" > gpurun_out/pytest_s2.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_s2.log
if grep -q "Memory access fault" gpurun_out/pytest_s2.log; then exit 1; fi
{
timeout -k 10 200 python tools/conv_bench.py --batch 1 --scale 2 --match "s2" --cfgs 0,11,12,13,18,19 || exit 1
timeout -k 10 200 python tools/conv_bench.py --batch 1 --scale 2 --match "conv_out" --cfgs 50,51 || exit 1
timeout -k 10 200 python tools/conv_bench.py --batch 1 --scale 1 --match "conv_out" --cfgs 50,51 || exit 1
timeout -k 10 200 python tools/conv_bench.py --batch 64 --scale 1 --match "conv_out" --cfgs 50,51 || exit 1
timeout -k 10 200 python tools/conv_bench.py --batch 32 --scale 2 --match "conv_out" --cfgs 50,51 || exit 1
timeout -k 10 200 python tools/conv_bench.py --batch 64 --scale 1 --match "s2" --cfgs 0,11,12,13,18,19 || exit 1
} > gpurun_out/conv_bench_latency_tiles.txt 2>&1
grep -v "sum over" gpurun_out/conv_bench_latency_tiles.txt
