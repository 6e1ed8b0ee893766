#!/bin/bash
# bf16x3 Winograd form: parity, three layers against the third f32 form, per-wave timeline
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/r03_ab7.sh || exit 1
export SISIC_LIB_PATH=$PWD/tools/bin/libsisic_hip_timing.so
timeout -k 10 200 python tools/bf3_timeline.py --cin 64 2>&1 | grep -v amdgpu > gpurun_out/bf3_timeline.txt || exit 1
timeout -k 10 200 python tools/bf3_timeline.py --cin 256 --cout 256 --hw 16 2>&1 | grep -v amdgpu >> gpurun_out/bf3_timeline.txt || exit 1
cat gpurun_out/bf3_timeline.txt
