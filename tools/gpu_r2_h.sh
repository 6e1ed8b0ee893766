#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for cfg in "1 128 1" "1 128 0" "1 64 1"; do timeout -k 10 200 python tools/latency_one.py $cfg 50 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/latency_detail.txt
bash tools/prof_script.sh lat128g tools/latency_one.py 1 128 1 50 | cut -c1-130 | head -14
bash tools/prof_script.sh trainp tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -24
