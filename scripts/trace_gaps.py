#!/usr/bin/env python3
"""Idle time between consecutive dispatches of a single-stream rocprofv3 kernel_trace.csv: per kernel, the mean
gap that FOLLOWS it, and busy/wall over the steady-state part of the run.  Dev tool."""
import collections, csv, glob, os, sys
root = sys.argv[1]
fs = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")[:44])
               for r in csv.DictReader(open(fs[0]))))
rows = rows[len(rows) // 2:]                                    # steady state: the second half of the run
busy = sum(e - s for s, e, _ in rows)
wall = rows[-1][1] - rows[0][0]
print("dispatches %d busy %.1f ms wall %.1f ms -> busy/wall %.3f" % (len(rows), busy / 1e6, wall / 1e6, busy / wall))
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[n0].append((s1 - e0) / 1e3)
print("%-44s %6s %10s %10s" % ("kernel (gap AFTER it)", "n", "mean_us", "median_us"))
for n, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    g.sort()
    print("%-44s %6d %10.2f %10.2f" % (n, len(g), sum(g) / len(g), g[len(g) // 2]))
