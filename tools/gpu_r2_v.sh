#!/bin/bash
# graph-replayed loop at batch 64 (SISIC_GRAPH=1) against launch-by-launch; training-step times and kernel statistics of the current library
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
mkdir -p gpurun_out
for g in 0 1; do
  SISIC_GRAPH=$g timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_graph$g.json 2> gpurun_out/bench_graph$g.log || exit 1
  echo "SISIC_GRAPH=$g"; grep timed gpurun_out/bench_graph$g.log
done
timeout -k 10 300 python tools/train_bench.py --steps 5 > gpurun_out/train_bench.txt 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/train_bench.txt
bash tools/prof_script.sh trainp tools/train_bench.py --batch 32 --size 64 --steps 3 | cut -c1-150 | head -30
