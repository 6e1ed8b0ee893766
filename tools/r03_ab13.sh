#!/bin/bash
# attention kernel: MFMA results in VGPRs, packed subtractions, permlane32_swap -- tests, then cfg2 / cfg4 before and after
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
for rep in 1 2; do
for v in persist4 new; do
  if [ $v = new ]; then unset SISIC_LIB_PATH; else export SISIC_LIB_PATH=$PWD/tools/bin/libsisic_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], {k: round(v,3) for k,v in j['roofline']['per_step_ms'].items()})"
  timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | head -1 | cut -c1-330
done
done
