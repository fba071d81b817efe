#!/bin/bash
# Ablation builds (wrong results, timing only): bench value per library, default split and unsplit
for L in "$@"; do
  for SP in 2 1; do
    v=$(MIDD_WS=0 MIDD_SPLIT=$SP MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'])" 2>&1 | tail -1)
    echo "$L split=$SP: $v"
  done
done
