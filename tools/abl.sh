#!/bin/bash
# Ablation builds (wrong results, timing only; hipcc -DC16_ABL=n on conv_mfma_f16x3.hip): bench value per
# library, default split and unsplit.  Wrong operands also change power draw and clocks: rows are upper bounds.
for L in "$@"; do
  for SP in 2 1; do
    v=$(MIDD_SPLIT=$SP MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'])" 2>&1 | tail -1)
    echo "$L split=$SP: $v"
  done
done
