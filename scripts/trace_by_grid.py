#!/usr/bin/env python3
"""Group a rocprofv3 kernel_trace.csv by (kernel, grid size): calls, mean duration, share.  Dev tool."""
import collections, csv, glob, os, sys
root = sys.argv[1]
fs = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")
    acc[(name[:48], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
print("%-48s %7s %7s %9s %9s %6s" % ("kernel", "wgs", "calls", "avg_us", "min_us", "%"))
for (name, wgs), v in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-48s %7d %7d %9.2f %9.2f %6.2f" % (name, wgs, len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
