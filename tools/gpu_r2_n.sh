#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log || { tail -5 gpurun_out/bench_default.log; exit 1; }
tail -4 gpurun_out/bench_default.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_s100_b.json 2> gpurun_out/bench_s100_b.log; tail -2 gpurun_out/bench_s100_b.log
