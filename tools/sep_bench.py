"""Separable objective (BASELINE config 5: N = 4096, D = 5) -- a few value and value+gradient evaluations, for rocprofv3.
    python tools/sep_bench.py [N] [M] [reps]        (NMGP_SEP=eig selects the reference's eigendecomposition formulation)
Prints one JSON line: ms per evaluation (wall, after warm-up) and the library's stage timers."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib, sim  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = sim.simulate_separable(N, M, 8)
pars = sim.perturb(d["pars_true"], 0.05, 0.4)
hv = [sim.HYPER_SEP[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma",
                                  "beta_tilde_sigma", "a", "b", "c")]
ctx = _lib.Context(0)
ctx.set_data(d["x"], d["Y"])
for _ in range(2):
    ctx.logpos_sep(pars, hv, True, True)
    ctx.logpos_sep(pars, hv, True, False)
res = {"N": N, "M": M, "formulation": os.environ.get("NMGP_SEP", "chol")}
for mode, key in ((False, "value"), (True, "value_grad")):
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, g = ctx.logpos_sep(pars, hv, True, mode)
    dt = (time.perf_counter() - t0) / reps
    pr = ctx.profile_read()
    ctx.profile_enable(False)
    res[key] = {"ms": 1e3 * dt, "stage_ms": {k: round(v[0] / max(v[1], 1) * (v[1] / reps), 4) for k, v in pr.items() if v[1]},
                "neglog": float(out[0])}
print(json.dumps(res))
