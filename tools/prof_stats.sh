#!/bin/bash
# rocprofv3 per-kernel statistics of a short bench.py run; run on the GPU box from the repo root:
#   bash tools/prof_stats.sh <tag> [bench args...]     -> gpurun_out/<tag>_kernel_stats.csv
set -e
tag=${1:-run}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- \
    python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-validate --profile-steps 1 "$@" > "$root/gpurun_out/${tag}_prof.log" 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
test -n "$f"
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
cut -c1-160 "$f" | sed -n 1,16p
