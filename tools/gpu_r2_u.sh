#!/bin/bash
# conv1x1.hip (tile_cfg 28 / 29): parity, per-layer time against the direct kernel's tilings, whole step
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv1x1 or partials or reference_layer" > gpurun_out/pytest_c1.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c1.log
if grep -q "Memory access fault" gpurun_out/pytest_c1.log; then exit 1; fi
grep -q "failed" gpurun_out/pytest_c1.log && exit 1
timeout -k 10 300 python tools/conv_bench.py --batch 64 --match "1x1" --cfgs 24,25,22,28,29,0 > gpurun_out/conv_bench_c1.txt 2>&1 || exit 1
grep -v "amdgpu.ids" gpurun_out/conv_bench_c1.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.log || exit 1
tail -2 gpurun_out/bench_c1.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c1.json'))
print(d['roofline']['per_step_ms'])
PY
