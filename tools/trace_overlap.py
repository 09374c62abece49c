"""Timeline summary of the LAST evaluation in a rocprofv3 kernel trace: per-kernel totals, wall time, and how much of the
wall had >= 2 kernels in flight (two-stream look-ahead)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_svc_finalize" in r["Kernel_Name"]]
ev = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(ev[0]["Start_Timestamp"])
tot = defaultdict(lambda: [0, 0.0])
pts = []
for r in ev:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    k = r["Kernel_Name"].split("(")[0][-40:]
    tot[k][0] += 1
    tot[k][1] += (b - a) * 1e-6
    pts.append((a, 1))
    pts.append((b, -1))
pts.sort()
depth, last, busy1, busy2 = 0, 0, 0, 0
for t, d in pts:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
wall = (max(int(r["End_Timestamp"]) for r in ev) - t0) * 1e-6
print("wall %.3f ms; >=1 kernel in flight %.3f ms; >=2 kernels in flight %.3f ms" % (wall, busy1 * 1e-6, busy2 * 1e-6))
for k, (c, ms) in sorted(tot.items(), key=lambda t: -t[1][1])[:8]:
    print("  %-42s x%4d %9.3f ms  (avg %.1f us)" % (k, c, ms, 1e3 * ms / c))
streams = defaultdict(float)
for r in ev:
    streams[(r.get("Queue_Id"), r.get("Stream_Id"))] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
print("per (queue, stream) busy ms:", dict(streams))
