#!/bin/bash
# measurement pass of the committed library (run through gpurun): full -m gpu suite (measured-error log), default bench (full T=1000),
# rocprofv3 kernel statistics of the same command, three PMC passes
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
mkdir -p gpurun_out
rm -f gpurun_out/errlog.txt
SISIC_TEST_ERRLOG=$R/gpurun_out/errlog.txt timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
# a failing, timed-out (124) or aborted (134) suite stops the pass: no numbers are published from a library that fails its tests
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/pytest_gpu.log; then tail -40 gpurun_out/pytest_gpu.log; exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log || { tail -5 gpurun_out/bench_default.log; exit 1; }
tail -4 gpurun_out/bench_default.log
bash tools/prof_stats.sh r03 | cut -c1-150 | head -14
cd $R && bash tools/prof_pmc.sh sq SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE | cut -c1-200 | head -8
cd $R && bash tools/prof_pmc.sh fetch FETCH_SIZE GRBM_GUI_ACTIVE | cut -c1-200 | head -6
cd $R && bash tools/prof_pmc.sh write WRITE_SIZE | cut -c1-200 | head -6
cd $R && python tools/make_pmc_summary.py gpurun_out gpurun_out/pmc_summary_r03.json
