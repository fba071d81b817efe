#!/bin/bash
# Cost of the batch-invariant plans, same box: tools/inv_bench.sh  (bench value default / invariant at batch 8 and 32, batch-1 latency)
for B in 8 32; do
  for inv in 0 1; do
    out=$(MIDD_BATCH_INVARIANT=$inv timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-iters 0 --latency-reps 2 --batch-per-gpu $B 2>/dev/null | tail -1)
    echo "B=$B invariant=$inv: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f img/s, batch-1 latency %.1f ms' % (d['value'], d.get('latency_batch1',{}).get('ms_per_image',0)))")"
  done
done
