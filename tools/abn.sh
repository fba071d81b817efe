#!/bin/bash
# Same-box comparison of N builds: tools/abn.sh rounds lib1.so lib2.so ...   (split and unsplit bench value per build)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    v=$(MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 --latency-reps 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    v1=$(MIDD_SPLIT=1 MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 --latency-reps 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    echo "$L split2=$v split1=$v1"
  done
done
