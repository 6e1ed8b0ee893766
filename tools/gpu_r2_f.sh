#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/prof_script.sh lat128 tools/latency_one.py 1 128 1 50 | cut -c1-150
tail -2 gpurun_out/lat128_prof.log
rm -f gpurun_out/errlog.txt
export SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog.txt
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "all rc=$?"; tail -6 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_s100_mix.json 2> gpurun_out/bench_s100_mix.log || echo "bench failed"
tail -3 gpurun_out/bench_s100_mix.log
