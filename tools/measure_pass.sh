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
bash tools/prof_stats.sh r04 | cut -c1-150 | head -14
cd $R && bash tools/prof_pmc.sh sq SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE | cut -c1-200 | head -8
cd $R && bash tools/prof_pmc.sh fetch FETCH_SIZE GRBM_GUI_ACTIVE | cut -c1-200 | head -6
cd $R && bash tools/prof_pmc.sh write WRITE_SIZE | cut -c1-200 | head -6
cd $R && python tools/make_pmc_summary.py gpurun_out gpurun_out/pmc_summary_r04.json
# per-layer: the bf16x3 kernels against the f32 forms they replace (B = 64), the per-wave timeline of the bf16x3 Winograd kernel,
# the training step and batch-1 latency with the round's final library
cd $R
{ echo "# python tools/conv_bench.py --cfgs 70|71,74: every stride-1 3x3 layer with whole 16x16-pixel tiles, third f32 Winograd form vs the bf16x3 form"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 71,74 --match "@64 gn" --iters 30 2>&1 | grep -v "amdgpu\|best cfg"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 70,74 --match "@32 gn" --iters 30 2>&1 | grep -v "amdgpu\|best cfg"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 70,74 --match "@16 gn" --iters 30 2>&1 | grep -v "amdgpu\|best cfg"
  echo "# python tools/conv_bench.py --cfgs 91,92 --match '@8 gn': the 8x8 level, K-split f32 form (two images per workgroup) vs K-split bf16x3 form (four)"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 91,92 --match "@8 gn" --iters 30 2>&1 | grep -v "amdgpu\|best cfg"
  echo "# python tools/conv_bench.py --cfgs 66,74 --match 'up ': the nearest-2x upsample convolutions, nine-position f32 form vs the bf16x3 form (16 positions)"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 66,74 --match "up " --iters 30 2>&1 | grep -v "amdgpu\|best cfg"
  echo "# python tools/conv_bench.py --cfgs 20,28,34,35 --match 1x1: the 1x1 layers, f32 pointwise kernel vs the bf16x3 forms (28 per wave, 34 staged, 35 K-split)"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 20,28,34,35 --match "1x1" --iters 30 2>&1 | grep -v "amdgpu\|best cfg"; } > gpurun_out/conv_bench_bf16x3.txt || exit 1
tail -3 gpurun_out/conv_bench_bf16x3.txt
if [ -f tools/bin/libsisic_hip_timing.so ]; then
    { SISIC_LIB_PATH=$R/tools/bin/libsisic_hip_timing.so timeout -k 10 200 python tools/bf3_timeline.py --cin 64 2>&1 | grep -v amdgpu
      SISIC_LIB_PATH=$R/tools/bin/libsisic_hip_timing.so timeout -k 10 200 python tools/bf3_timeline.py --cin 128 --cout 128 --hw 32 2>&1 | grep -v amdgpu
      SISIC_LIB_PATH=$R/tools/bin/libsisic_hip_timing.so timeout -k 10 200 python tools/bf3_timeline.py --cin 256 --cout 256 --hw 16 2>&1 | grep -v amdgpu; } > gpurun_out/bf3_timeline.txt || exit 1
fi
timeout -k 10 400 python tools/train_bench.py > gpurun_out/train_bench.txt 2> gpurun_out/train_bench.log || { tail -5 gpurun_out/train_bench.log; exit 1; }
cat gpurun_out/train_bench.txt
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/configs_4_5.jsonl 2> gpurun_out/configs_4_5.log || { tail -5 gpurun_out/configs_4_5.log; exit 1; }
cut -c1-300 gpurun_out/configs_4_5.jsonl
