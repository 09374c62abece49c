"""MAP point of the bench subject: the nonseparable model at N = 2048, D = 3, data seed 2222 (bench.py's rank-0 subject; 2222 is the
reference's single-subject seed, SIM_code/sim.py:359), hyper-parameters of Nonseparable_model.py:274-275.

    python tools/make_map_point.py [iterations, default 1000] [out, default gpurun_out/map_N2048_M3_seed2222.npz]     (on an MI355X)

The MAP loop of Nonseparable_model.py:147-210 -- Adam, lr 0.2, N_opt = 1000 -- run by drivers.BatchedMAP (parameters, gradients and
moments resident in HBM), started from the same smooth perturbation of the generating parameters as bench.py's chain 0.  The
result (P = 14,337 doubles + the 1000-step objective history) is committed as tests/golden/map_N2048_M3_seed2222.npz: bench.py
starts its HMC chains there, as Nonseparable_model.py:228-231 starts the reference's sampler from MAP.dat."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nonstationary_multivariate_gaussian_process_amd import sim  # noqa: E402
from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedMAP  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "map_N2048_M3_seed2222.npz")
    N, M, seed = 2048, 3, 2222
    d = sim.simulate_nonseparable(N, M, seed=seed)
    p0 = sim.perturb(d["pars_true"], 0.05, 0.7)                 # bench.chain_parameters(..)[0]
    bm = BatchedMAP(d["x"][None], d["Y"][None], sim.HYPER_SVC, p0[None], lr=0.2)
    t0 = time.time()
    pars, hist, alive = bm.run(iters)
    dt = time.time() - t0
    assert alive[0] and np.all(np.isfinite(hist))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, pars_map=pars[0], pars0=p0, target_value_hist=hist[:, 0], N=N, M=M, seed=seed, iterations=iters,
                        lr=0.2, seconds=dt)
    print("MAP: %d Adam iterations in %.1f s; log posterior %.6f -> %.6f (truth: see bench); wrote %s" % (
        iters, dt, hist[0, 0], hist[-1, 0], out))


if __name__ == "__main__":
    main()
