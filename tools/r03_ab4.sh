#!/bin/bash
# third Winograd form: phase order of the waves and s_setprio around the MFMAs (SISIC_WINO_FLAGS)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for rep in 1 2; do
for fl in 0 1 2 4 5 6; do
  echo "=== rep $rep SISIC_WINO_FLAGS=$fl"
  SISIC_WINO_FLAGS=$fl timeout -k 10 300 python tools/conv_bench.py --cfgs 71 --match "64 @64 gn" --iters 40 2>&1 | grep " 71 " || exit 1
  SISIC_WINO_FLAGS=$fl timeout -k 10 300 python tools/conv_bench.py --cfgs 70 --match "128 @32 gn" --iters 40 2>&1 | grep " 70 " || exit 1
  SISIC_WINO_FLAGS=$fl timeout -k 10 300 python tools/conv_bench.py --cfgs 70 --match "256 @16 gn" --iters 40 2>&1 | grep " 70 " || exit 1
done
done > gpurun_out/ab_flags.txt
python - <<'PY'
import re,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
var=None
for ln in open('gpurun_out/ab_flags.txt'):
    m=re.match(r"=== rep (\d) SISIC_WINO_FLAGS=(\d)",ln)
    if m: var=int(m.group(2)); continue
    m=re.match(r"(.{34})\s+\d+\s+(\d+)\s+([\d.]+)",ln)
    if m: d[m.group(1).strip()][var].append(float(m.group(3)))
vs=[0,1,2,4,5,6]
print(f"{'layer':34s} " + " ".join(f"flags{v:d}      " for v in vs))
for k,v in d.items():
    print(f"{k:34s} " + " ".join("/".join(f"{x:.1f}" for x in v[i]) for i in vs))
PY
