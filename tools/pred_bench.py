"""Deterministic prediction at the reference's grid (Nonseparable_model.py:333: 201 points) -- accuracy against the committed
reference golden (N = 512) and wall time at the headline size (N = 2048, D = 3).
    python tools/pred_bench.py [N for the timing, default 2048] [grid points, default 201] [repetitions, default 5] [N_sep D_sep]
Prints one JSON line.  Under `rocprofv3 --kernel-trace --stats` the per-kernel table gives k_svc_crosscov's duration; its
algorithmic traffic is the cross-covariance it writes, 8 n S M bytes (n = N M), quoted in the line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nonstationary_multivariate_gaussian_process_amd import _lib, sim  # noqa: E402

SVC_KEYS = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")


def errs(mean, var, Ls, ref, Lref):
    rv = ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2
    return {"mean_max_abs": float(np.max(np.abs(mean - ref[:, 1]))),
            "mean_max_rel_over_scale": float(np.max(np.abs(mean - ref[:, 1])) / np.max(np.abs(ref[:, 1]))),
            "mean_worst_allclose_ratio": float(np.max(np.abs(mean - ref[:, 1]) / (1e-7 + 1e-5 * np.abs(ref[:, 1])))),
            "var_max_rel": float(np.max(np.abs(var - rv) / rv)),
            "Lstar_max_abs": float(np.max(np.abs(Ls - Lref)))}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 201
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    M = 3
    rec = {}
    gpath = os.path.join(ROOT, "tests", "golden", "pred_N512_M3_grid201.npz")
    c = _lib.Context(0)
    if os.path.exists(gpath):
        g = np.load(gpath)
        c.set_data(g["x"], g["Y"])
        mean, var, Ls = c.predict_svc(g["svc_pars"], g["svc_hyper"], g["grids"])
        rec["parity_N512_grid201"] = errs(mean, var, Ls, g["svc_pct"], g["svc_Lstar"])
        if "svc_rough_pars" in g:
            mean, var, Ls = c.predict_svc(g["svc_rough_pars"], g["svc_hyper"], g["svc_rough_grids"])
            rec["parity_N512_rough"] = errs(mean, var, Ls, g["svc_rough_pct"], g["svc_rough_Lstar"])
    d = sim.simulate_nonseparable(N, M, seed=2222)
    hv = np.array([sim.HYPER_SVC[k] for k in SVC_KEYS])
    xs = np.linspace(0.0, 1.0, S)
    c.set_data(d["x"], d["Y"])
    p = d["pars_true"]
    c.predict_svc(p, hv, xs)             # prior factors, buffers
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        mean, var, Ls = c.predict_svc(p, hv, xs)
        ts.append(time.perf_counter() - t0)
    n = N * M
    rec["timing"] = {"what": "nmgp_predict_svc, host pointers in and out, N=%d, D=%d, %d grid points" % (N, M, S),
                     "ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * float(np.min(ts)),
                     "crosscov_algorithmic_bytes": 8.0 * n * S * M, "mean_finite": bool(np.all(np.isfinite(mean))),
                     "var_positive": bool(np.all(var > 0))}
    # separable (config 5's shape when N = 4096, D = 5) and stationary predictors on the same grid: one eigendecomposition of K_x
    # for all grid points (the reference: one per grid point, prediction.py:381-382)
    if len(sys.argv) > 4:
        Ns, Ms = int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 5
        ds = sim.simulate_separable(Ns, Ms, seed=8)
        hs = np.array([sim.HYPER_SEP[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma",
                                                   "beta_tilde_sigma", "a", "b", "c")])
        c.set_data(ds["x"], ds["Y"])
        c.predict_sep(ds["pars_true"], hs, xs)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            m2, v2 = c.predict_sep(ds["pars_true"], hs, xs)
            ts.append(time.perf_counter() - t0)
        rec["timing_sep"] = {"what": "nmgp_predict_sep, N=%d, D=%d, %d grid points" % (Ns, Ms, S), "ms_median": 1e3 * float(np.median(ts)),
                             "finite": bool(np.all(np.isfinite(m2)) and np.all(v2 > 0))}
        dt_ = sim.simulate_stationary(Ns, Ms, seed=8)
        c.set_data(dt_["x"], dt_["Y"])
        c.predict_sta(dt_["pars_true"], xs)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            m3, v3 = c.predict_sta(dt_["pars_true"], xs)
            ts.append(time.perf_counter() - t0)
        rec["timing_sta"] = {"what": "nmgp_predict_sta, N=%d, D=%d, %d grid points" % (Ns, Ms, S), "ms_median": 1e3 * float(np.median(ts)),
                             "finite": bool(np.all(np.isfinite(m3)) and np.all(v3 > 0))}
    print(json.dumps(rec))
    c.close()


if __name__ == "__main__":
    main()
