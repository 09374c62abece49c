"""Where the gradient's distance to the CPU oracle comes from: likelihood part vs GP-prior part (python tests/diag_grad_split.py on the GPU box; lives under tests/ because it uses the CPU oracle as the checker)."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib, sim
from oracle import nmgp_oracle as O
keys = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")
for N in (1024, 2048):
    d = sim.simulate_nonseparable(N, 3, 11)
    pars = sim.perturb(d["pars_true"], 0.05, 0.2)
    hv = [sim.HYPER_SVC[k] for k in keys]
    ctx = _lib.Context(0)
    ctx.set_data(d["x"], d["Y"])
    res = {}
    for prior in (False, True):
        out, g = ctx.logpos_svc(pars, hv, prior=prior, want_grad=True)
        ref, gr = O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **sim.HYPER_SVC, verbose=True, grad=True, Prior=prior)
        res[prior] = (g, gr)
        print(N, "prior", prior, "vec_relerr", np.linalg.norm(g - gr) / np.linalg.norm(gr), "norm", np.linalg.norm(gr))
    gp, grp = res[True][0] - res[False][0], res[True][1] - res[False][1]
    print(N, "prior part alone: relerr", np.linalg.norm(gp - grp) / np.linalg.norm(grp), "norm", np.linalg.norm(grp))
    # which blocks
    T = 6
    for name, sl in (("tilde_l", slice(0, N)), ("uL", slice(N, N + N * T))):
        print("   ", name, np.linalg.norm((gp - grp)[sl]) / np.linalg.norm(grp[sl]))
