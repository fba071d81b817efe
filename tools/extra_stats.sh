#!/bin/bash
# Kernel stats of the other BASELINE shapes (run on the GPU box): tools/extra_stats.sh TAG -> gpurun_out/prof_TAG/kernel_stats_{b32,512}.csv, per_op_b1_256.txt
TAG=${1:-r03}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kb32 -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --latency-reps 0 --batch-per-gpu 32 > $OUT/kb32.log 2>&1
cp $(find $OUT/kb32 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_b32.csv; rm -rf $OUT/kb32
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/k512 -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --latency-reps 0 --size 512 > $OUT/k512.log 2>&1
cp $(find $OUT/k512 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_512.csv; rm -rf $OUT/k512
python tools/per_op_profile.py 1 256 > $OUT/per_op_b1_256.txt 2>&1; head -2 $OUT/per_op_b1_256.txt | tail -1
head -4 $OUT/kernel_stats_b32.csv | cut -c1-140; head -4 $OUT/kernel_stats_512.csv | cut -c1-140
