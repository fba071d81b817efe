#!/bin/bash
# Ablation builds of the WS kernel (wrong results, timing only): unsplit bench value per library
for L in medical-image-denoising-using-diffusion_amd/libmidd.so libmidd_abl1.so libmidd_abl2.so libmidd_abl3.so libmidd_abl4.so; do
  v=$(MIDD_SPLIT=1 MIDD_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-iters 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d['kernels'] if 'ws' in x['name']]; print('%.2f img/s  ws kernel %.1f ms' % (d['value'], k[0]['ms'] if k else -1))" 2>&1 | tail -1)
  echo "$L: $v"
done
