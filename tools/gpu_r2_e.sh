#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 300 python tools/conv_bench.py --cfgs 66,68,69 --match " gn" > gpurun_out/conv_bench_wide.txt 2>&1
grep -v "^1x1\|conv_out\|@8 " gpurun_out/conv_bench_wide.txt | tail -60
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -k "train or adam or winograd or partials or overfits or scaler or gradients" > gpurun_out/pytest_train.log 2>&1
echo "train rc=$?"; tail -6 gpurun_out/pytest_train.log
timeout -k 10 300 python tools/latency_b1.py > gpurun_out/latency_b1.txt 2>&1; cat gpurun_out/latency_b1.txt | tail -12
timeout -k 10 300 python tools/train_bench.py --steps 5 > gpurun_out/train_bench.txt 2>&1; tail -3 gpurun_out/train_bench.txt
