#!/bin/bash
# Single-GPU bench lines of the other BASELINE.json configurations (run on the GPU box): tools/bench_configs.sh TAG
# -> gpurun_out/prof_TAG/bench_{default,c3_b32_n100,c4shard_b32,c5shard_512}.json
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 500 python3 bench.py "$@" > $OUT/bench_$name.log 2>&1; grep '^{"metric' $OUT/bench_$name.log | tail -1 > $OUT/bench_$name.json; python3 -c "import json,sys; d=json.load(open('$OUT/bench_$name.json')); print('$name', round(d['value'],2), d['unit'], round(d['ms_per_step'],1), 'ms/step', d['roofline']['traffic'])"; }
run default
run c3_b32_n100 --noise-steps 100 --inference-steps 100 --batch-per-gpu 32 --steps 2 --warmup 1 --cpu-iters 0 --latency-reps 0
run c4shard_b32 --batch-per-gpu 32 --steps 2 --warmup 1 --cpu-iters 0 --latency-reps 0
run c5shard_512 --size 512 --batch-per-gpu 8 --steps 2 --warmup 1 --cpu-iters 0 --latency-reps 0
