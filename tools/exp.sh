#!/bin/bash
# Same-box experiments on the current build: tools/exp.sh "ENV1=a ENV2=b" "ENV3=c" ...  (one bench run per argument)
for E in "$@"; do
  v=$(env $E timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f img/s  b1 %.1f ms' % (d['value'], d.get('latency_batch1',{}).get('ms_per_image',0)))" 2>&1 | tail -1)
  echo "[$E] $v"
done
