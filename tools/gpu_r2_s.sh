#!/bin/bash
# where the in-place K-split reduction loses its time: rocprofv3 kernel durations of one layer under tile_cfg 76 and 78
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for cfg in 76 78; do
  rm -rf /tmp/prof_k$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k$cfg -o k$cfg -- \
      python3 "$root/tools/conv_bench.py" --batch 1 --scale 2 --match "64->64 @64 gn+res" --cfgs $cfg > "$root/gpurun_out/k${cfg}_prof.log" 2>&1 || exit 1
  f=$(find /tmp/prof_k$cfg -name "*kernel_stats.csv" | head -1)
  test -n "$f" || exit 1
  cp "$f" "$root/gpurun_out/k${cfg}_kernel_stats.csv"
  echo "cfg $cfg"; cut -c1-170 "$f" | sed -n 1,6p
done
