#!/bin/bash
# lean pointwise 1x1 kernel (tile_cfg 20) against the generic tilings, per layer
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv1x1 or finalize or partials" > gpurun_out/ab5_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/ab5_pytest.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/ab5_pytest.log; exit 1; fi
for rep in 1 2; do
timeout -k 10 300 python tools/conv_bench.py --cfgs 20,24,25 --match "1x1" --iters 40 2>&1 | grep -v amdgpu || exit 1
done > gpurun_out/ab_pointwise.txt
cat gpurun_out/ab_pointwise.txt
