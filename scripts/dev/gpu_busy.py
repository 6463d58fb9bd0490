"""dev: GPU busy time per training step from a rocprofv3 kernel trace of scripts/micro/host_profile.py (65 RPN steps)."""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(fs[0])))
steps = int(sys.argv[2])
# skip set-up: take the last 60 % of the dispatches
rows = rows[int(len(rows) * 0.4):]
busy = 0; cur_s, cur_e = rows[0]
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print("dispatches %d, span %.1f ms, GPU busy (union of kernel intervals) %.1f ms = %.3f of the span; sum of kernel durations %.1f ms" % (
    len(rows), span / 1e6, busy / 1e6, busy / span, sum(e - s for s, e in rows) / 1e6))
