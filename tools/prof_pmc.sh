#!/bin/bash
# One rocprofv3 counter pass over a two-step bench.py run (counters in their own run, kernel-trace only):
#   bash tools/prof_pmc.sh <tag> <COUNTER> [COUNTER...]   -> gpurun_out/<tag>_counters.csv (per-kernel averages)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o $tag -- \
    python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-validate --profile-steps 0 > "$root/gpurun_out/${tag}_pmc.log" 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
test -n "$f"
python3 - "$f" "$root/gpurun_out/${tag}_counters.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as out:
    for k in sorted(acc, key=lambda k: -len(n[k])):
        line = f"{k} | dispatches {len(n[k])} | " + " ".join(f"{c}={v / len(n[k]):.4g}" for c, v in sorted(acc[k].items()))
        out.write(line + "\n")
        print(line[:230])
PY
