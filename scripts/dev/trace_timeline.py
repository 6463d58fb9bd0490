#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace as a timeline: start (us from the step's first kernel), duration, gap to the previous
kernel's end on the same stream, stream, kernel.  The step before the last optimiser launch.   usage: trace_timeline.py <dir>"""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for r in csv.DictReader(open(fs[0])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:60], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_sgd_momentum_v4") or r[2].startswith("k_adam")]
steps = [marks[0]]
for m in marks[1:]:
    if rows[m][0] - rows[steps[-1]][0] > 300000:
        steps.append(m)
lo, hi = steps[-3], steps[-2]
t0 = rows[lo][0]
last = {}
print("step span %.1f us" % ((rows[hi][0] - rows[lo][0]) / 1e3))
for s, e, name, st, q in rows[lo:hi + 1]:
    gap = (s - last[st]) / 1e3 if st in last else 0.0
    last[st] = e
    print("%8.1f %7.1f  gap %7.1f  s%-3s q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, st, q, name))
