#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per dispatch of kernels matching a pattern."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "conv_igemm")
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-36s n=%3d mean=%.4g" % (k, len(v), sum(v) / len(v)))
