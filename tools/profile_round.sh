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
# Traffic: ONE half-batch program (batch 4, the launches the two-stream run issues) alone on one stream.  Counters are sampled
# device-wide around a dispatch: with two streams a kernel's window also holds the co-resident kernel's bytes (round 3: in_conv's
# 50.3 MB output read 60.7 MB that way, 50.6 MB alone).
CMD="python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --latency-reps 0 --inference-steps 2 --batch-per-gpu 4"
MIDD_SPLIT=1 MIDD_PLAN_AS_SIDE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- $CMD > $OUT/pf.log 2>&1
MIDD_SPLIT=1 MIDD_PLAN_AS_SIDE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- $CMD > $OUT/pw.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv" | head -1) $(find $OUT/pw -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json "MIDD_SPLIT=1 MIDD_PLAN_AS_SIDE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- $CMD  (one half-batch program of the default run, alone: per-dispatch counters are device-wide, a co-resident kernel's bytes would be counted too)"
rm -rf $OUT/kt $OUT/pf $OUT/pw
cat $OUT/bench.json
