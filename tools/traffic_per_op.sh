#!/bin/bash
# L2-side traffic of every launch of ONE forward (run on the GPU box): tools/traffic_per_op.sh NAME [B]
# One program on one stream (MIDD_SPLIT=1), planned as a sub-batch program of the default two-stream run (MIDD_PLAN_AS_SIDE), so
# the dispatch order is the op order and the launches are the default run's; the last complete forward of the run is listed.
NAME=$1; B=${2:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tpo_$NAME; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MIDD_SPLIT=1 MIDD_PLAN_AS_SIDE=1
[ -n "$3" ] && export MIDD_LIBRARY=$GRAFT_REPO_ROOT/$3
CMD="python3 bench.py --steps 1 --warmup 0 --cpu-iters 0 --latency-reps 0 --inference-steps 2 --batch-per-gpu $B"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- $CMD > $OUT/pf.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- $CMD > $OUT/pw.log 2>&1
python3 - <<PY
import csv, glob
def load(d, counter):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "midd::" in (r.get("Kernel_Name") or "")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].replace("void midd::", "").split("(")[0], float(r["Counter_Value"])) for r in rows]
fe, wr = load("pf", "FETCH_SIZE"), load("pw", "WRITE_SIZE")
n = 73
starts = [i for i, (k, _) in enumerate(fe) if k.startswith("in_conv") and i + n <= len(fe)]      # the last complete forward
fe, wr = fe[starts[-1]:starts[-1] + n], wr[starts[-1]:starts[-1] + n]
with open("$OUT.txt", "w") as o:
    for i, ((k, f), (k2, w)) in enumerate(zip(fe, wr)):
        o.write("op%03d %-62s fetch %7.1f MB  write %6.1f MB\n" % (i, k[:62], 2 * f * 1024 / 1e6, w * 1024 / 1e6))
print(open("$OUT.txt").read())
PY
rm -rf $OUT
