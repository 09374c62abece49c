"""MFMA utilisation of k_syrk_lower from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES).

usage: python tools/pmc_mfma.py <counter_collection.csv> [out.json] [end marker, default k_svc_finalize; k_svc_grad_final for value+gradient]
Per launch of the LAST evaluation: duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the MFMA-busy share
of the SIMD cycles, SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * clock)."""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
by = defaultdict(dict)
meta = {}
for r in rows:
    d = int(r["Dispatch_Id"])
    by[d][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[d] = (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
ids = sorted(meta)
MARK = sys.argv[3] if len(sys.argv) > 3 else "k_svc_finalize"
fin = [d for d in ids if MARK in meta[d][0]]
lo = fin[-2] if len(fin) > 1 else ids[0]
out = []
for d in ids:
    if d <= lo or d > fin[-1] or "k_syrk_lower" not in meta[d][0]:
        continue
    name, t0, t1 = meta[d]
    dur = (t1 - t0) * 1e-9
    c = by[d]
    clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / dur if dur > 0 else 0.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    # the counter is summed over the SIMDs of the chip: 256 CUs x 4
    util = busy / (1024.0 * dur * clk) if clk > 0 else 0.0
    out.append({"dispatch": d, "us": dur * 1e6, "clock_ghz": clk * 1e-9, "mfma_busy_share": util})
big = sorted(out, key=lambda o: -o["us"])[:6]
tot = sum(o["us"] for o in out)
avg_util = sum(o["mfma_busy_share"] * o["us"] for o in out) / tot if tot else 0.0
avg_clk = sum(o["clock_ghz"] * o["us"] for o in out) / tot if tot else 0.0
res = {"what": "k_syrk_lower launches of the last evaluation (end marker %s): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock), "
               "clock = GRBM_GUI_ACTIVE / 8 XCDs / duration" % MARK,
       "launches": len(out), "total_us": tot, "time_weighted_mfma_busy_share": avg_util, "time_weighted_clock_ghz": avg_clk,
       "clock_limited_peak_tflops": 78.6 * avg_clk / 2.4, "largest_launches": big}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
