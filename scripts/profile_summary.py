#!/usr/bin/env python3
"""Condense a profile_round.sh output directory into the text summary committed under profiles/."""
import collections, csv, glob, json, os, sys
import re
root = sys.argv[1]


def bench_kernel_name(k):
    """rocprof kernel name -> the name bench.py / ops.CONV_KERNEL_NAMES use for the same instantiation."""
    m = re.search(r"k_conv_igemm_f32_v2<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>", k)
    if m:
        tm, tn, var, wm, wn, sk, stem = m.groups()
        args = [tm, tn] + ([var] if var != "0" or (wm, wn) != ("2", "2") else []) + ([wm, wn] if (wm, wn) != ("2", "2") else [])
        return "k_conv_igemm_f32_v2<%s>" % ",".join(args) + (" split-K" if sk == "true" else "") + (" stem" if stem == "true" else "")
    m = re.search(r"k_conv_igemm_f32_sk<(\d+), (\d+)>", k)
    if m:
        return "k_conv_igemm_f32_sk<%s,%s>" % m.groups()
    m = re.search(r"k_conv_igemm_x6<(\d+), (\d+), (\d+), (\d+)(, (true|false))?>", k)
    if m:
        return "k_conv_igemm_x6<%s,%s,%s,%s>" % m.groups()[:4] + (" split-K" if m.group(6) == "true" else "")
    if "k_conv_igemm_x6_db" in k:
        return "k_conv_igemm_x6_db"
    m = re.search(r"k_conv_igemm_h3_db<(\d+), (\d+), (\d+), (\d+)(, (true|false))?(, \d+)?>", k)      # (.., planes in, ring loop 0 / 1): the two plane-reading instantiations are ONE line
    if m:
        return "k_conv_igemm_h3_db<%s,%s,%s,%s%s>" % (m.groups()[:4] + (",planes" if m.group(6) == "true" else "",))
    m = re.search(r"k_conv_igemm_h3<(\d+), (\d+), (\d+), (\d+)(, (true|false))?>", k)
    if m:
        return "k_conv_igemm_h3<%s,%s,%s,%s>" % m.groups()[:4] + (" split-K" if m.group(6) == "true" else "")
    m = re.search(r"k_conv_igemm_bf16<", k)
    if m:
        return "k_conv_igemm_bf16"
    return k


print("== bench lines")
for f in ("bench_default.json", "bench_streams1.json"):
    p = os.path.join(root, f)
    if os.path.exists(p):
        print(f, open(p).read().strip())
for tag in ("trace_streams1", "trace_default"):
    fs = sorted(glob.glob(os.path.join(root, tag, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize, reverse=True)
    if not fs:
        continue                                    # (several processes leave one file each: the largest is bench.py's own)
    print("\n== rocprofv3 --kernel-trace --stats:", tag)
    print("%-78s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in csv.DictReader(open(fs[0])):
        if float(r["Percentage"]) < 0.05:
            continue
        print("%-78s %7s %12.1f %10.2f %6.2f" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
# Launches that ran ALONE on the device (no other dispatch overlaps them in time): with several images in flight a
# kernel's duration in the trace includes the time it shares the chip with other streams' kernels, so the figure
# comparable with bench.py's roofline (each distinct launch re-issued back to back on one stream) is the average
# over the isolated dispatches -- which are mostly those very re-issues.
for tag in ("trace_streams1", "trace_default"):
    fs = sorted(glob.glob(os.path.join(root, tag, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize, reverse=True)
    if not fs:
        continue
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fs[0]))]
    rows.sort()
    iso = collections.defaultdict(list)
    every = collections.defaultdict(list)
    max_end = 0
    for i, (st, en, name) in enumerate(rows):
        alone = st >= max_end and (i + 1 == len(rows) or rows[i + 1][0] >= en)
        max_end = max(max_end, en)
        if "conv_igemm" in name:
            every[name].append(en - st)
            if alone:
                iso[name].append(en - st)
    print("\n== conv dispatches that overlap no other dispatch:", tag)
    print("%-78s %7s %10s | %7s %10s" % ("kernel", "alone", "avg_us", "all", "avg_us"))
    for name in sorted(every, key=lambda n: -sum(every[n])):
        a, e = iso.get(name, []), every[name]
        print("%-78s %7d %10.2f | %7d %10.2f" % (name.split("(")[0][:78], len(a), (sum(a) / len(a) / 1e3) if a else float("nan"), len(e), sum(e) / len(e) / 1e3))
    groups = collections.defaultdict(list)                     # bench.py's kernel names that cover several instantiations: one more row each
    for name in every:
        groups[bench_kernel_name(name.split("(")[0])].append(name)
    for g, members in groups.items():
        if len(members) > 1:
            a = [d for m in members for d in iso.get(m, [])]
            e = [d for m in members for d in every[m]]
            print("%-78s %7d %10.2f | %7d %10.2f   <- bench.py's roofline.avg_launch_us is this average" % (("= " + g)[:78], len(a), (sum(a) / len(a) / 1e3) if a else float("nan"), len(e), sum(e) / len(e) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm" in k or "nms" in k or "topk_rank" in k or "roi_fwd" in k:
            # bench.py reports ONE bf16 conv kernel (all instantiations, averaged over an image's launches): same key here
            key = "frcnn::k_conv_igemm_bf16<all instantiations>" if "k_conv_igemm_bf16<" in k else k.split("(")[0][:70]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
# bench.py's name for a kernel may cover several instantiations (the plane-reading 256x128 kernel: double-buffer and ring loops): their
# dispatches are averaged TOGETHER for traffic.json, and printed as one more entry beside the instantiations' own
canon = collections.defaultdict(lambda: collections.defaultdict(list))
for k in acc:
    for n, v in acc[k].items():
        canon[bench_kernel_name(k)][n] += v
for name in canon:
    members = [k for k in acc if bench_kernel_name(k) == name]
    if len(members) > 1:
        acc[name + "  (= " + " + ".join(m.split("frcnn::")[-1] for m in members) + ")"] = canon[name]
traffic = {}
if acc:
    print("\n== PMC means per dispatch (rocprofv3 --pmc, separate passes, eager single stream)")
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k, "dispatches/pass:", len(next(iter(acc[k].values()))))
    for n in sorted(c):
        print("    %-34s %.4g" % (n, c[n]))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM)
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        name = bench_kernel_name(k.split("  (= ")[0])
        if "  (= " in k or name not in traffic:                 # (the combined entry wins over a member's own)
            traffic[name] = {"hbm_bytes_per_launch": round(hbm), "fetch_kb_raw": c["FETCH_SIZE"], "write_kb": c["WRITE_SIZE"]}
        print("    -> HBM-side bytes per launch (2*FETCH+WRITE)*1024 = %.3g" % hbm)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        print("    -> matrix pipe busy = MFMA_BUSY/1024 SIMDs / (GUI_ACTIVE/8 XCDs) = %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)))
    if "TCC_HIT_sum" in c:
        print("    -> L2 hit rate = %.3f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
# whole-pass totals per PMC directory (all kernels): HBM-side bytes of everything a pass launched
print("\n== PMC totals per pass directory (all kernels summed; KB as the counters report them)")
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    tot = collections.defaultdict(float)
    nd = 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            nd += 1
    log = d + ".log"
    note = open(log).read().strip().splitlines()[-1][:120] if os.path.exists(log) else ""
    print(os.path.basename(d), {k: "%.4g" % v for k, v in tot.items()}, "rows", nd, "|", note)
