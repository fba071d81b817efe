#!/bin/bash
# HBM-side traffic per launch of one build (run on the GPU box): tools/traffic_only.sh NAME [lib.so]  -> gpurun_out/traffic_NAME.json
NAME=$1; LIB=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$NAME; mkdir -p $OUT
[ -n "$LIB" ] && export MIDD_LIBRARY=$GRAFT_REPO_ROOT/$LIB
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --latency-reps 0 --inference-steps 2"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- $CMD > $OUT/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- $CMD > $OUT/pw.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv" | head -1) $(find $OUT/pw -name "*counter_collection.csv" | head -1) $OUT.json "traffic_only $NAME"
rm -rf $OUT
python3 - <<PY
import json
d=json.load(open("$OUT.json"))
for k,v in d["kernels"].items():
    if "conv_mfma" in k: print("$NAME", k[6:70], "%.1f MB fetch %.1f MB write" % (v["fetch_bytes_corrected"]/1e6, v["write_bytes"]/1e6), v["launches_sampled"])
PY
