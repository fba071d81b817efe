#!/bin/bash
# Profiles of the default bench for one round (run on the GPU box): tools/profile_round.sh TAG
# -> gpurun_out/prof_TAG/{kernel_stats.csv, bench_under_rocprof.json, pmc_traffic.json, bench.json}
set -e
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 --latency-reps 0 > $OUT/kt.log 2>&1
grep '^{"metric' $OUT/kt.log | tail -1 > $OUT/bench_under_rocprof.json
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
CMD="python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --latency-reps 0 --inference-steps 2"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- $CMD > $OUT/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- $CMD > $OUT/pw.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv" | head -1) $(find $OUT/pw -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- $CMD  (default split: two half-batches of 4)"
rm -rf $OUT/kt $OUT/pf $OUT/pw
cat $OUT/bench.json
