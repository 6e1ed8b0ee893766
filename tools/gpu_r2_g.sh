#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sampler.py -q -x > gpurun_out/pytest_part.log 2>&1 || { tail -20 gpurun_out/pytest_part.log; exit 1; }
tail -3 gpurun_out/pytest_part.log
timeout -k 10 300 python tools/latency_b1.py > gpurun_out/latency_b1.txt 2>&1 || { tail -5 gpurun_out/latency_b1.txt; exit 1; }
if grep -q "Memory access fault" gpurun_out/latency_b1.txt; then tail -5 gpurun_out/latency_b1.txt; exit 1; fi
tail -12 gpurun_out/latency_b1.txt
