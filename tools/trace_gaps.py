"""Stream occupancy from a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Per queue: busy time and the gaps between consecutive kernels; over all queues: time with 0 / 1 / 2+ kernels resident."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3   # kernels [lo_f, hi_f) of the trace by launch order: bench.py runs
hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7   # warm-up, the timed calls, then the per-launch-event leg
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows if r["Kernel_Name"].startswith(("midd::", "void midd::", "_ZN4midd"))]
ev.sort()
ev = ev[int(lo_f * len(ev)):int(hi_f * len(ev))]
t0, t1 = ev[0][0], max(e[1] for e in ev)
byq = collections.defaultdict(list)
for e in ev: byq[e[2]].append(e)
print(f"window {1e-6 * (t1 - t0):.1f} ms, {len(ev)} kernels on {len(byq)} queues")
for q, l in byq.items():
    busy = sum(e[1] - e[0] for e in l)
    gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
    gaps = [g for g in gaps if g < 1e6]        # drop inter-call pauses
    gs = sorted(gaps)
    print(f"queue {q}: {len(l)} kernels, busy {1e-6 * busy:.1f} ms ({100.0 * busy / (t1 - t0):.1f} %), gap median {gs[len(gs)//2] / 1e3:.2f} us, mean {sum(gs) / len(gs) / 1e3:.2f} us, sum {1e-6 * sum(gs):.1f} ms")
pts = []
for s, e, _, _ in ev: pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter(); cur = 0; last = pts[0][0]
for t, d in pts:
    hist[min(cur, 3)] += t - last; last = t; cur += d
tot = sum(hist.values())
print("resident kernels: " + ", ".join(f"{k}: {100.0 * v / tot:.1f} %" for k, v in sorted(hist.items())))
