#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/errlog.txt
export SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd or partials or fuzz" > gpurun_out/pytest_wide.log 2>&1
echo "wide rc=$?"; tail -8 gpurun_out/pytest_wide.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "all rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_s100_wide.json 2> gpurun_out/bench_s100_wide.log || echo "bench failed"
tail -3 gpurun_out/bench_s100_wide.log
SISIC_WINO_WIDE=0 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_s100_nowide.json 2> gpurun_out/bench_s100_nowide.log || echo "bench failed"
tail -2 gpurun_out/bench_s100_nowide.log
