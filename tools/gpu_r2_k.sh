#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 300 python tools/conv_bench.py --cfgs 0,24,25,28,29 --match 1x1 2>&1 | grep -v "amdgpu.ids\|Traceback\|File\|main()\|print\|ZeroDiv" | tee gpurun_out/conv1x1_occ.txt
bash tools/prof_script.sh trainp2 tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -30
