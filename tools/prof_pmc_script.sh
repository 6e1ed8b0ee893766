#!/bin/bash
# One rocprofv3 counter pass over any python script of this repo, kept per dispatch (kernel, grid, counters):
#   bash tools/prof_pmc_script.sh <tag> "<COUNTER COUNTER ...>" tools/bench_configs.py --only 4 --steps 2
# -> gpurun_out/<tag>_dispatches.csv
set -e
tag=$1; counters=$2; script=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d /tmp/pmcs_$tag -o $tag -- \
    python3 "$root/$script" "$@" > "$root/gpurun_out/${tag}_pmc.log" 2>&1
f=$(find /tmp/pmcs_$tag -name "*counter_collection.csv" | head -1)
test -n "$f"
python3 - "$f" "$root/gpurun_out/${tag}_dispatches.csv" <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:80], r.get("Grid_Size", ""))
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
with open(sys.argv[2], "w") as out:
    for (d, k, g), c in rows.items():
        out.write(f"{d},{k},{g}," + ",".join(f"{n}={v:.6g}" for n, v in sorted(c.items())) + "\n")
print(len(rows), "dispatches")
PY
