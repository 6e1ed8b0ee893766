#!/bin/bash
# pointwise bf16x3 kernel: what bounds it?  diagnostic builds without the filter loads / without the x loads (results wrong)
for v in hip pwb_NO_FILTER_LOADS pwb_NO_X_LOADS; do
  if [ $v = hip ]; then unset SISIC_LIB_PATH; else export SISIC_LIB_PATH=$PWD/tools/bin/libsisic_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/conv_bench.py --cfgs 28 --iters 30 --match "1x1" 2>&1 | grep -v "^sum\|amdgpu.ids\|best cfg"
done
