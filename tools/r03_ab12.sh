#!/bin/bash
# persistent workgroups in the bf16x3 Winograd kernel: library before (one item per workgroup) vs the persistent stages, same box
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd or tail" 2>&1 | tail -3
for rep in 1 2 3; do
for v in shipped persist3 persist4; do
  export SISIC_LIB_PATH=$PWD/tools/bin/libsisic_$v.so
  echo "== $v"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], {k: round(v,3) for k,v in j['roofline']['per_step_ms'].items()})"
done
done
