"""Separable objective, B chains per launch sequence (nmgp_sep_batch_eval) at BASELINE config 5's shape (N = 4096, D = 5):
    python tools/sep_batch_bench.py [N] [M] [chains ...]
Prints one JSON line per chain count: evaluations / s and ms per chain for value and value+gradient, and the fraction of the FP64
matrix roofline (M N^3 / 3 flop per value evaluation, M N^3 per value+gradient evaluation)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib, sim  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
chains = [int(v) for v in sys.argv[3:]] or [1, 4, 8, 16]
d = sim.simulate_separable(N, M, 8)
hv = [sim.HYPER_SEP[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma",
                                  "beta_tilde_sigma", "a", "b", "c")]
ctx = _lib.Context(0)
ctx.set_data(d["x"], d["Y"])
PEAK = 78.6e12
for B in chains:
    pars = np.stack([sim.perturb(d["pars_true"], 0.05, 0.4 + 0.1 * k) for k in range(B)])
    rec = {"N": N, "M": M, "chains": B}
    for mode, key, flop in ((False, "value", M * float(N) ** 3 / 3.0), (True, "value_grad", M * float(N) ** 3)):
        reps = max(2, 24 // B)
        for _ in range(2):
            ctx.sep_batch_eval(pars, hv, True, mode)
        ctx.profile_enable(True)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            out, g, st = ctx.sep_batch_eval(pars, hv, True, mode)
        dt = (time.perf_counter() - t0) / reps
        pr = ctx.profile_read()
        ctx.profile_enable(False)
        assert np.all(st == 0)
        rec[key] = {"ms_per_batch": 1e3 * dt, "ms_per_chain": 1e3 * dt / B, "evals_per_s": B / dt,
                    "roofline_frac_end_to_end": B * flop / dt / PEAK,
                    "stage_ms_per_batch": {k: round(v[0] / reps, 3) for k, v in pr.items() if v[1]}}
    print(json.dumps(rec), flush=True)
