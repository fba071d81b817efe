"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel name."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        name = name.replace("void midd::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(name, r["Counter_Name"])] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("GRBM_GUI_ACTIVE", 0)))
for n in names[:14]:
    c = agg[n]
    k = max(cnt[(n, x)] for x in c)
    line = f"{n[:60]:60s} n={k:5d} " + " ".join(f"{x.replace('SQ_','')}={v/k:.3g}" for x, v in sorted(c.items()))
    print(line)
