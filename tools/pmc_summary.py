"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into HBM bytes per batched evaluation.

usage: python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [note] [end marker]
(end marker: the kernel that closes an evaluation -- k_svc_finalize, the default, for value evaluations, k_svc_grad_final for
value+gradient ones, whose inverse SYRK and adjoint pass run after k_svc_finalize)

Scope: the LAST complete evaluation in each trace (first k_svc_cov dispatch after the previous k_svc_finalize ..
the final k_svc_finalize).  Counter values are KB (rocprofv3 derived counters).  Per MI355X_MICROARCH.md (HBM section)
FETCH_SIZE on gfx950 reports half the bytes of wide coalesced streaming reads, so the read side is doubled; WRITE_SIZE is
taken as is.  The SYRK kernel's C-tile loads are 8 B/lane (uncalibrated width), so its doubled read figure is an upper
bound.
"""
import csv
import json
import sys
from collections import OrderedDict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


MARKER = "k_svc_finalize"


def last_eval(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    fin = [i for i, r in enumerate(rows) if MARKER in r["Kernel_Name"]]
    if not fin:
        raise SystemExit("no %s dispatch in %s" % (MARKER, path))
    end = fin[-1]
    begin = fin[-2] + 1 if len(fin) > 1 else 0
    cov = [i for i in range(begin, end) if "k_svc_prep" in rows[i]["Kernel_Name"] or "k_svc_cov" in rows[i]["Kernel_Name"]]
    if cov:
        begin = cov[0]
    per = OrderedDict()
    for r in rows[begin:end + 1]:
        k = short(r["Kernel_Name"])
        d = per.setdefault(k, {"dispatches": 0, "sum_KB": 0.0})
        d["dispatches"] += 1
        d["sum_KB"] += float(r["Counter_Value"])
    return per


def main():
    fetch, write, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    global MARKER
    if len(sys.argv) > 5:
        MARKER = sys.argv[5]
    f = last_eval(fetch)
    w = last_eval(write)
    tot_f = sum(v["sum_KB"] for v in f.values())
    tot_w = sum(v["sum_KB"] for v in w.values())
    syrk_f = sum(v["sum_KB"] for k, v in f.items() if "k_syrk_lower" in k)
    syrk_w = sum(v["sum_KB"] for k, v in w.items() if "k_syrk_lower" in k)
    syrk_n = sum(v["dispatches"] for k, v in f.items() if "k_syrk_lower" in k)
    res = {
        "summary": {
            "scope": "the last complete batched evaluation in the trace (k_svc_prep/k_svc_cov .. %s)" % MARKER,
            "note": note,
            "FETCH_SIZE_KB": tot_f, "WRITE_SIZE_KB": tot_w,
            "hbm_bytes_gfx950_corrected": 1024.0 * (2.0 * tot_f + tot_w),
            "k_syrk_lower": {"launches": syrk_n, "FETCH_SIZE_KB": syrk_f, "WRITE_SIZE_KB": syrk_w,
                             "hbm_bytes_gfx950_corrected": 1024.0 * (2.0 * syrk_f + syrk_w),
                             "hbm_bytes_per_launch_avg": 1024.0 * (2.0 * syrk_f + syrk_w) / max(syrk_n, 1)},
        },
        "FETCH_SIZE": f, "WRITE_SIZE": w,
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["summary"], indent=1))


if __name__ == "__main__":
    main()
