#!/usr/bin/env python3
"""Timeline of the last iteration of graph_event_probe.py from a rocprofv3 kernel trace: per kernel start / duration / queue."""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for r in csv.DictReader(open(fs[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
# last burst: kernels after the last gap > 5 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 5_000_000:
        cut = i
t0 = rows[cut][0]
for s, e, n, st, q in rows[cut:]:
    tag = "mul" if "Mul" in n or "mul" in n else "add"
    print("%8.1f %6.1f s%-3s q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, q, tag))
