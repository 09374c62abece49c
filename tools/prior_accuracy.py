"""Which Cholesky gets the ill-conditioned GP-prior terms closer to the reference (golden vectors)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden
from nonstationary_multivariate_gaussian_process_amd import _lib
for name in ["svc_sim_N1024_M3_base", "svc_sim_N2048_M3_base", "svc_rngfree_N1024_M3", "svc_sim_N128_M3_dist"]:
    g = golden(name)
    for algo in ("custom", "rocsolver"):
        os.environ["NMGP_CHOL"] = algo
        c = _lib.Context(0)
        os.environ.pop("NMGP_CHOL")
        c.set_data(g["x"], g["Y"])
        out, grad = c.logpos_svc(g["pars"], g["hyper"], True, True)
        rel = np.abs(out - g["out"]) / np.abs(g["out"])
        print("%-26s %-9s rel err vs reference: NegLog %.1e loglik %.1e lp_l %.1e lp_uL %.1e | grad %.1e" % (
            name, algo, rel[0], rel[1], rel[2], rel[3], np.linalg.norm(grad - g["grad"]) / np.linalg.norm(g["grad"])))
        c.close()
