#!/bin/bash
# Same-box sweep of one environment knob: tools/sweep_env.sh VAR v1 v2 ...   (bench value per setting, two rounds)
VAR=$1; shift
for r in 1 2; do
  for v in "$@"; do
    val=$(env $VAR=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 --latency-reps 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    echo "$VAR=$v: $val"
  done
done
