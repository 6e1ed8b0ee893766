#!/bin/bash
# rocprofv3 per-kernel statistics of tools/train_bench.py (batch 32, 64x64); run on the GPU box from the repo root:
#   bash tools/prof_train.sh <tag> [ENV=VALUE ...]     -> gpurun_out/<tag>_train_kernel_stats.csv, gpurun_out/<tag>_train.log
set -e
tag=${1:-train}; shift || true
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- \
    python3 "$root/tools/train_bench.py" --batch 32 --size 64 --steps 5 > "$root/gpurun_out/${tag}_train.log" 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
test -n "$f"
cp "$f" "$root/gpurun_out/${tag}_train_kernel_stats.csv"
grep '^{' "$root/gpurun_out/${tag}_train.log"
cut -c1-150 "$f" | sed -n 1,12p
