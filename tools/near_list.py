import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_svc_finalize" in r["Kernel_Name"]]
ev = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(ev[0]["Start_Timestamp"])
for r in ev:
    n = r["Kernel_Name"]
    if "syrk" in n or "potf2" in n:
        print("%8.1f %8.1f %s grid=%s" % ((int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, n.split("(")[0][-20:], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
