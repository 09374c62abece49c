"""Per-launch-class efficiency of k_syrk_lower from a rocprofv3 kernel trace of `bench.py` (N=2048, D=3 by default).

usage: python tools/syrk_classes.py <kernel_trace.csv> [n] [nb1] [batch] [grad: 0 | 1]
grad = 1: a value+gradient evaluation (marker k_svc_grad_final; the factorisation carries the pad row and the n rows of L^-T;
the inverse SYRK launch is listed separately).  The tiles of L^-T rows start their k-loop at their first non-zero k-panel
(syrk_tile_body: tri_row0 / tri_k0, an even number of 16-column panels per 128-row tile): the flop figures count what the tiles
EXECUTE -- `launch_flop` replays that rule -- so no class can exceed the matrix peak (round 3's table counted the skipped zero
panels as flop and printed 88.7 / 93.6 TFLOP/s for the K = 2048 classes).
Replays the factorisation's launch schedule (recursive-halving panels + trailing updates) to attach (mrows, ncols, K)
to the SYRK launches of the LAST evaluation in the trace, then prints time, TFLOP/s and algorithmic TB/s per class.
"""
import csv
import os
import sys
from collections import defaultdict


LEAF = os.environ.get("NMGP_CHOL_LEAF", "1") != "0"


def schedule(n, nb1, extra=1, xtri=0):
    """(rows, ncols, K) of every update launch of one factorisation.  xtri > 0: the gradient evaluation, whose `xtri` rows of
    L^-T ride below the matrix -- the first c1 of them take part once the factorisation has reached column c1."""
    sched = []          # (rows, ncols, K, c0, c1): the update of columns [c1, c1 + ncols) with the panel [c0, c1), K = c1 - c0

    def act(c1):
        return n + extra + min(c1, xtri) - c1

    def rec(c0, w):
        # (leaf: the library's default schedule ends the recursion at 128 columns -- k_panel_step<1>, <2> fold the K = 64
        # update into the second solve; NMGP_CHOL_LEAF=0 in the environment of this tool restores the 64-column recursion)
        if w <= 64 or (LEAF and w == 128):
            return
        h = ((w // 2 + 63) // 64) * 64
        if h >= w:
            h = ((w - 1) // 64) * 64
        rec(c0, h)
        c1 = c0 + h
        below = act(c1)
        if below > 0:
            sched.append((below, w - h, h, c0, c1))
        rec(c1, w - h)

    for c0 in range(0, n, nb1):
        w1 = min(nb1, n - c0)
        rec(c0, w1)
        c1 = c0 + w1
        if c1 < n:
            sched.append((act(c1), n - c1, w1, c0, c1))
    return sched


def launch_flop(m, nc, K, c0, c1, n, extra, xtri):
    """Executed flop of one update launch per matrix: 2 K_t per lower-trapezoid element, where K_t = K minus the leading zero
    k-panels a 128-row tile of L^-T rows skips (nmgp_chol.hip, syrk_tile_body: rows >= tri_row0 = n + extra - c1 of the launch are
    rows of L^-T, A[i, k] == 0 for k < (i - tri_row0) - tri_k0 with tri_k0 = c0; kt0 = even part of (row0 - tri_row0 - tri_k0) / 16)."""
    if not xtri:
        return 2.0 * K * (nc * m - 0.5 * nc * (nc - 1))
    tri_row0 = n + extra - c1
    gy = (nc + 127) // 128
    fl = 0.0
    row0 = 0
    while row0 < m:
        rows = min(128, m - row0)
        bi = row0 // 128
        # elements (i >= j) of this tile row
        if bi < gy:
            full_cols = 128 * bi
            w = min(128, nc - full_cols)
            elems = rows * full_cols + (w * (w + 1) / 2.0 if rows >= w else rows * (rows + 1) / 2.0) + max(0, rows - w) * w
        else:
            elems = rows * nc
        kt0 = 0
        if rows == 128 and row0 >= tri_row0:             # (the leftover rows below the last full tile ride in the diagonal tiles: full K)
            z = (row0 - tri_row0 - c0) // 16
            kt0 = (z & ~1) if z > 0 else 0
        fl += 2.0 * (K - 16 * kt0) * elems
        row0 += 128
    return fl


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
    nb1 = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    grad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if ("k_svc_grad_final" if grad else "k_svc_finalize") in r["Kernel_Name"]]
    ev = rows[idx[-2] + 1: idx[-1] + 1]
    sched = schedule(n, nb1, 2 if grad else 1, n if grad else 0)
    sy = [r for r in ev if "k_syrk" in r["Kernel_Name"]]
    if grad:
        # the last k_syrk_lower launch of a value+gradient evaluation is -Sigma^-1 = -X X^T (K = n, triangular operand: n^3/3 flop)
        inv = sy[-1]
        sy = sy[:-1]
        d = (int(inv["End_Timestamp"]) - int(inv["Start_Timestamp"])) * 1e-9
        fl = batch * float(n) ** 3 / 3.0
        print("inverse SYRK (K = n = %d, triangular operand)  time=%8.3f ms  %6.1f TF/s on n^3/3 flop per matrix" % (n, d * 1e3, fl / d / 1e12))
    if len(sy) != len(sched):
        raise SystemExit("launch count mismatch: trace %d, schedule %d" % (len(sy), len(sched)))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for r, (m, nc, K, c0, c1) in zip(sy, sched):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        elems = nc * m - 0.5 * nc * (nc - 1)
        a = agg[(K, nc)]
        a[0] += 1
        a[1] += d
        a[2] += launch_flop(m, nc, K, c0, c1, n, 2 if grad else 1, n if grad else 0) * batch
        a[3] += 8 * batch * (2 * elems + m * K)
    tot = 0.0
    for k in sorted(agg):
        c, d, fl, by = agg[k]
        tot += d
        print("K=%4d ncols=%5d launches=%3d time=%7.3f ms  %6.1f TF/s (executed flop)  %5.2f TB/s (algorithmic)  %5.1f flop/B" % (
            k[0], k[1], c, d * 1e3, fl / d / 1e12, by / d / 1e12, fl / by))
    print("k_syrk_lower total %.3f ms" % (tot * 1e3))
    oth = defaultdict(float)
    for r in ev:
        oth[r["Kernel_Name"].split("(")[0][-48:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for k, v in sorted(oth.items(), key=lambda t: -t[1])[:8]:
        print("  %-50s %8.3f ms" % (k, v))
    print("wall of the evaluation: %.3f ms" % ((int(ev[-1]["End_Timestamp"]) - int(ev[0]["Start_Timestamp"])) * 1e-6))


if __name__ == "__main__":
    main()
