"""Per-kernel summary of the LAST complete evaluation in a rocprofv3 kernel trace (kernel_trace.csv):
python tools/trace_summary.py <kernel_trace.csv> [end-marker kernel substring, default k_svc_finalize]
Prints per kernel: launches, total / mean / min / max duration, and the gaps between consecutive kernels."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_svc_finalize"
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    ev = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) > 1 else rows
    agg = defaultdict(list)
    gaps = 0.0
    prev_end = None
    for r in ev:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "").split("(")[0][-40:]
        agg[name].append((e - s) * 1e-3)
        if prev_end is not None and s > prev_end:
            gaps += (s - prev_end) * 1e-3
        prev_end = max(prev_end or e, e)
    wall = (int(ev[-1]["End_Timestamp"]) - int(ev[0]["Start_Timestamp"])) * 1e-3
    print("%-42s %6s %10s %8s %8s %8s" % ("kernel", "n", "total_us", "mean", "min", "max"))
    for k, v in sorted(agg.items(), key=lambda t: -sum(t[1])):
        print("%-42s %6d %10.1f %8.2f %8.2f %8.2f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    print("wall %.1f us, kernel sum %.1f us, idle gaps %.1f us, %d launches" % (wall, sum(sum(v) for v in agg.values()), gaps,
                                                                              len(ev)))


if __name__ == "__main__":
    main()
