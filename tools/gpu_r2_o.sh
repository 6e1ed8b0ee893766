#!/bin/bash
# A/B of the shortcut fork (SISIC_FORK): sampler + unet tests, bench, single-image latency
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sampler.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/pytest_fork.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_fork.log
if grep -q "Memory access fault" gpurun_out/pytest_fork.log; then exit 1; fi
for f in 0 1; do
  SISIC_FORK=$f timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_fork$f.json 2> gpurun_out/bench_fork$f.log || exit 1
  echo "fork=$f"; grep "timed" gpurun_out/bench_fork$f.log
  SISIC_FORK=$f timeout -k 10 300 python tools/latency_one.py 1 128 1 > gpurun_out/latency_fork$f.txt 2>&1 || exit 1
  cat gpurun_out/latency_fork$f.txt
  SISIC_FORK=$f timeout -k 10 300 python tools/latency_one.py 1 64 1 >> gpurun_out/latency_fork$f.txt 2>&1 || exit 1
  tail -2 gpurun_out/latency_fork$f.txt
done
