#!/bin/bash
# round-2 first GPU pass: full -m gpu suite with the measured-error log, a short bench line, host-RNG ceiling, MFMA/VALU probe
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -f gpurun_out/errlog.txt
SISIC_TEST_ERRLOG=$PWD/gpurun_out/errlog.txt timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_s100.json 2> gpurun_out/bench_s100.log || echo "bench failed"
tail -3 gpurun_out/bench_s100.log
timeout -k 10 120 python tools/noise_producers.py --ranks 8 --workers 16 --seconds 4 > gpurun_out/noise_producers.json 2>&1
cat gpurun_out/noise_producers.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/probe && timeout -k 10 120 /tmp/probe > gpurun_out/mfma_valu_probe.txt 2>&1
tail -30 gpurun_out/mfma_valu_probe.txt
