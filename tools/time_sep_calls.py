"""Per-call wall time of the separable objective (value / value+gradient), to spot one-off costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib, sim
ctx = _lib.Context(0)
N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 5
d = sim.simulate_separable(N, M, 5)
pars = sim.perturb(d["pars_true"], 0.05, 0.4)
hv = [sim.HYPER_SEP[k] for k in ["mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma",
                                  "beta_tilde_sigma", "a", "b", "c"]]
ctx.set_data(d["x"], d["Y"])
ctx.profile_enable(True)
for mode in [True, False, False, False, True, True, False, False]:
    ctx.profile_reset()
    t0 = time.perf_counter()
    out = ctx.logpos_sep(pars, hv, True, mode)
    t1 = time.perf_counter()
    pr = ctx.profile_read()
    print("grad" if mode else "value", "%.2f ms" % ((t1 - t0) * 1e3), {k: round(v[0], 2) for k, v in pr.items() if v[1]}, flush=True)
