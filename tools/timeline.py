"""Kernel timeline of the LAST complete evaluation in a rocprofv3 kernel trace: start (us from the evaluation's first
kernel), duration, gap to the previous kernel's end (negative = overlap with another stream), kernel, grid.
usage: python tools/timeline.py <dir or kernel_trace.csv> [end-marker kernel substring, default k_svc_finalize]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_svc_finalize"
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    ev = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) > 1 else rows
    t0 = int(ev[0]["Start_Timestamp"])
    pe = t0
    for r in ev:
        n = r["Kernel_Name"].split("(")[0][-24:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.1f dur %7.1f gap %7.1f %-24s grid=%s" % ((s - t0) * 1e-3, (e - s) * 1e-3, (s - pe) * 1e-3, n,
                                                          r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
        pe = e


if __name__ == "__main__":
    main()
