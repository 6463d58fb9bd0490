#!/usr/bin/env python3
"""What ONE training step is made of: a rocprofv3 kernel trace of `bench_train.py --only rpn|det --steps N` cut at the optimiser
launches (one k_sgd_momentum_v4 per step), the last third of the steps averaged: per (kernel, grid) us per step on each stream,
the union of kernel intervals, the idle time of the main stream.  Dev tool.   usage: trace_step.py <dir> [rows]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
fs = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for r in csv.DictReader(open(fs[0])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")
    wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:52], wgs, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_sgd_momentum_v4") or r[2].startswith("k_adam")]
# the optimiser runs once per step (two launches in some configs: keep the first of each close pair)
steps = [marks[0]]
for m in marks[1:]:
    if rows[m][0] - rows[steps[-1]][0] > 300000: steps.append(m)
n = len(steps) - 1
lo = steps[n - max(1, n // 3)]; hi = steps[n]
sel = rows[lo:hi]; k = max(1, n // 3)
span = (sel[-1][1] - sel[0][0]) / 1e3 / k
busy = 0; cs, ce = sel[0][0], sel[0][1]
for s, e, *_ in sel[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, name, wgs, st in sel:
    a = acc[(name, wgs, st)]; a[0] += 1; a[1] += (e - s) / 1e3
print("steps averaged %d; step span %.1f us; union of kernel intervals %.1f us (%.0f %%); sum of kernel durations %.1f us; dispatches per step %.0f" % (
    k, span, busy / 1e3 / k, 100 * busy / 1e3 / k / span, sum(a[1] for a in acc.values()) / k, len(sel) / k))
print("%-52s %6s %7s %7s %9s %9s" % ("kernel", "wgs", "stream", "calls", "avg_us", "us/step"))
for (name, wgs, st), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-52s %6d %7s %7.1f %9.2f %9.1f" % (name, wgs, st, c / k, t / c, t / k))
