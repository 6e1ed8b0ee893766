#!/bin/bash
# round-2 second GPU pass: training tests first (new code), then the whole suite with the measured-error log
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/errlog.txt
export SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog.txt
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/pytest_train.log 2>&1
echo "train rc=$?"
tail -30 gpurun_out/pytest_train.log
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_train.py > gpurun_out/pytest_gpu.log 2>&1
echo "all rc=$?"
tail -15 gpurun_out/pytest_gpu.log
