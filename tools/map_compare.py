"""Where the reference's MAP recipe ends and what it costs to reach the mode (nonseparable model, N = 2048, D = 3, bench subject):

    python tools/map_compare.py [out.json]                                                            (on an MI355X)

(a) the MAP loop of Nonseparable_model.py:147-210 as written -- Adam, lr 0.2, 1000 iterations (drivers.BatchedMAP, device-resident);
(b) drivers.polish_map from THE SAME start point, no Adam at all: L-BFGS in the prior-whitened coordinates, preconditioned by the
    low-rank likelihood curvature (the prior-factor metric), metric rebuilt every round;
(c) (a) followed by (b).
Prints one JSON document: log posterior reached, seconds, gradient evaluations."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonstationary_multivariate_gaussian_process_amd import _lib, drivers, sim  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "map_compare.json")
    N, M, seed = 2048, 3, 2222
    d = sim.simulate_nonseparable(N, M, seed=seed)
    h = sim.HYPER_SVC
    hv = np.array([h[k] for k in drivers.SVC_HYPER_KEYS])
    p0 = sim.perturb(d["pars_true"], 0.05, 0.7)
    c = _lib.default_context()
    c.set_data(d["x"], d["Y"])

    def logpost(p):
        return -float(c.logpos_svc(p, hv, True, False)[0][0])
    rec = {"subject": "nonseparable N = %d, D = %d, data seed %d; start = the generating parameters under a smooth 5 %% perturbation "
                      "(bench.py's chain 0)" % (N, M, seed),
           "log_posterior": {"start": logpost(p0), "generating_parameters": logpost(d["pars_true"])}}
    log = []
    t0 = time.time()
    qb, nl, gn, nev = drivers.polish_map(d["x"], d["Y"], h, p0, maxiter=600, rounds=8, verbose=log.append)
    rec["preconditioned_lbfgs_from_the_start_point"] = {"log_posterior": -nl, "seconds": time.time() - t0, "gradient_evaluations": nev,
                                                        "whitened_gradient_norm": gn, "rounds": log}
    print(json.dumps(rec["preconditioned_lbfgs_from_the_start_point"])[:400], flush=True)
    t0 = time.time()
    pars, hist, alive = drivers.BatchedMAP(d["x"][None], d["Y"][None], h, p0[None], lr=0.2).run(1000)
    ta = time.time() - t0
    rec["adam_1000_as_the_reference_runs_it"] = {"log_posterior": float(hist[-1, 0]), "log_posterior_after_100_300_1000": [float(hist[k, 0]) for k in (99, 299, 999)],
                                                 "seconds": ta, "gradient_evaluations": 1000, "reference": "Nonseparable_model.py:147-210 (Adam, lr 0.2)"}
    log2 = []
    t0 = time.time()
    qc, nl2, gn2, nev2 = drivers.polish_map(d["x"], d["Y"], h, pars[0], maxiter=300, verbose=log2.append)
    rec["adam_then_preconditioned_lbfgs"] = {"log_posterior": -nl2, "seconds": ta + time.time() - t0, "gradient_evaluations": 1000 + nev2,
                                             "whitened_gradient_norm": gn2, "rounds": log2,
                                             "rms_distance_between_the_two_modes": float(np.sqrt(np.mean((qb - qc) ** 2)))}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
