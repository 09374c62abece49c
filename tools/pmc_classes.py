"""HBM traffic of k_syrk_lower per launch class (K, ncols) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py.

usage: python tools/pmc_classes.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [n] [nb1] [batch] [extra | grad]
("grad": a value+gradient evaluation -- marker k_svc_grad_final, pad row + n rows of L^-T below the matrix, and the inverse SYRK
launch, the last k_syrk_lower of the evaluation, reported as its own class)

Replays the batched factorisation's launch schedule (recursive-halving panels + trailing updates, as tools/syrk_classes.py)
to attach (rows, ncols, K) to the SYRK dispatches of the LAST evaluation in each pass, then reports per class the measured
bytes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, counter unit KB) next to the algorithmic bytes
(read + write of the updated trapezoid + the panel once) -- i.e. WHERE the excess traffic of the update kernel lands."""
import csv
import json
import sys
from collections import OrderedDict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from syrk_classes import launch_flop, schedule  # noqa: E402


def last_eval_syrk(path, marker="k_svc_finalize"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    fin = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    ev = rows[fin[-2] + 1: fin[-1] + 1] if len(fin) > 1 else rows
    return [float(r["Counter_Value"]) for r in ev if "k_syrk_lower" in r["Kernel_Name"]]


def main():
    fetch, write, out = sys.argv[1:4]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 6144
    nb1 = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 128
    grad = len(sys.argv) > 7 and sys.argv[7] == "grad"
    extra = 2 if grad else (int(sys.argv[7]) if len(sys.argv) > 7 else 1)
    sched = schedule(n, nb1, extra, n if grad else 0)
    marker = "k_svc_grad_final" if grad else "k_svc_finalize"
    f, w = last_eval_syrk(fetch, marker), last_eval_syrk(write, marker)
    inv = None
    if grad:
        inv = (f.pop(), w.pop())
    if len(f) != len(sched) or len(w) != len(sched):
        raise SystemExit("launch count mismatch: fetch %d, write %d, schedule %d" % (len(f), len(w), len(sched)))
    cls = OrderedDict()
    for (m, nc, K, c0, c1), fk, wk in zip(sched, f, w):
        elems = nc * m - 0.5 * nc * (nc - 1)
        c = cls.setdefault("K=%d" % K, {"launches": 0, "measured_bytes": 0.0, "algorithmic_bytes": 0.0, "flop": 0.0})
        c["launches"] += 1
        c["measured_bytes"] += 1024.0 * (2.0 * fk + wk)
        c["algorithmic_bytes"] += 8.0 * batch * (2.0 * elems + m * K)
        c["flop"] += launch_flop(m, nc, K, c0, c1, n, extra, n if grad else 0) * batch        # EXECUTED flop (zero k-panels skipped)
    if inv is not None:
        # -Sigma^-1 = -X X^T: reads the upper-triangular X once (8 n^2 / 2), writes both triangles of the result (8 n^2)
        cls["inverse (K=n)"] = {"launches": 1, "measured_bytes": 1024.0 * (2.0 * inv[0] + inv[1]),
                                "algorithmic_bytes": 8.0 * batch * 1.5 * n * n, "flop": batch * float(n) ** 3 / 3.0}
    tot_m = sum(c["measured_bytes"] for c in cls.values())
    tot_a = sum(c["algorithmic_bytes"] for c in cls.values())
    for c in cls.values():
        c["ratio"] = c["measured_bytes"] / c["algorithmic_bytes"]
        c["flop_per_measured_byte"] = c["flop"] / c["measured_bytes"]
    res = {"n": n, "nb1": nb1, "batch": batch, "classes": cls, "total_measured_bytes": tot_m,
           "total_algorithmic_bytes": tot_a, "ratio": tot_m / tot_a,
           "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KB counters; k_syrk_lower launches of the last batched evaluation"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
