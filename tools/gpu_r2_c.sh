#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 300 python tools/debug_adam.py > gpurun_out/debug_adam.log 2>&1; tail -12 gpurun_out/debug_adam.log
rm -f gpurun_out/errlog.txt
export SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "all rc=$?"
tail -12 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_s100_global.json 2> gpurun_out/bench_s100_global.log || echo "bench failed"
tail -3 gpurun_out/bench_s100_global.log
