"""Acceptance / energy error of BatchedHMC from the committed MAP point for several leapfrog step sizes (N = 2048, D = 3):
    python tools/hmc_steps.py [chains, default 16] [samples, default 4] [step sizes ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonstationary_multivariate_gaussian_process_amd import drivers, sim  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = [float(v) for v in sys.argv[3:]] or [1e-4, 5e-5, 3e-5, 2e-5]
    g = np.load(os.path.join(ROOT, "tests", "golden", "map_N2048_M3_seed2222.npz"))
    d = sim.simulate_nonseparable(2048, 3, seed=2222)
    q0 = np.repeat(g["pars_map"][None], B, 0)
    for eps in steps:
        h = drivers.BatchedHMC(d["x"], d["Y"], sim.HYPER_SVC, q0, step_size=eps, num_steps_in_leap=20, seed=1)
        s, info = h.run(S)
        ee = info["energy_error"]
        print(json.dumps({"step_size": eps, "chains": B, "samples": S, "accept_rate_mean": float(info["accept_rate"].mean()),
                          "median_dH": float(np.nanmedian(ee)), "median_abs_dH": float(np.nanmedian(np.abs(ee))),
                          "rms_move": float(np.sqrt(np.mean((s[-1] - q0) ** 2)))}), flush=True)


if __name__ == "__main__":
    main()
