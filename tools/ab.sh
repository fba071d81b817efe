#!/bin/bash
# Same-box A/B of two builds: tools/ab.sh libA.so libB.so [rounds]   (devices differ by several % -- never compare across boxes)
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    v=$(MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    v1=$(MIDD_SPLIT=1 MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    echo "$L split2=$v split1=$v1"
  done
done
