#!/bin/bash
# copies what tools/measure_pass.sh left under gpurun_out/ into profiles/<round>/ under the names bench.py and the docs read
#   bash tools/publish_pass.sh r03 last_library
set -e
rnd=${1:?round}; tag=${2:?tag of the bench line}
d=profiles/$rnd; g=gpurun_out
cp $g/bench_default.json $d/bench_default_T1000_$tag.json
cp $g/bench_default.log $d/bench_default_T1000_$tag.log
cp $g/${rnd}_kernel_stats.csv $d/bench_steps20_kernel_stats.csv
cp $g/pmc_summary_${rnd}.json $d/pmc_summary.json
for t in sq fetch write; do test -s $g/${t}_counters.csv; cp $g/${t}_counters.csv $d/pmc_${t}_per_kernel.txt; done
sort -g -r $g/errlog.txt | head -400 > $d/test_errors.txt
cp $g/conv_bench_bf16x3.txt $g/bf3_timeline.txt $g/train_bench.txt $g/configs_4_5.jsonl $d/
git status --short $d
