#!/bin/bash
# round-3 A/B pass on the GPU box: kernel parity first, then per-layer timings of the second (72/73) and third (70/71) forms
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
true
rc=0
tail -5 gpurun_out/ab_pytest.log
if [ $rc -ne 0 ]; then tail -60 gpurun_out/ab_pytest.log; exit 1; fi
timeout -k 10 400 python tools/conv_bench.py --cfgs 73,71 --match "@64" --iters 30 > gpurun_out/ab_conv64.txt 2>&1 || { tail -20 gpurun_out/ab_conv64.txt; exit 1; }
cat gpurun_out/ab_conv64.txt
timeout -k 10 400 python tools/conv_bench.py --cfgs 72,70 --match "@32" --iters 30 > gpurun_out/ab_conv32.txt 2>&1 || { tail -20 gpurun_out/ab_conv32.txt; exit 1; }
cat gpurun_out/ab_conv32.txt
timeout -k 10 400 python tools/conv_bench.py --cfgs 72,70 --match "@16" --iters 30 > gpurun_out/ab_conv16.txt 2>&1 || { tail -20 gpurun_out/ab_conv16.txt; exit 1; }
cat gpurun_out/ab_conv16.txt
