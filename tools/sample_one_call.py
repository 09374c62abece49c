"""drivers.sample_nonseparable -- the whole recipe behind one call -- on ANOTHER subject than the bench's (data seed 2223 by default)
at the headline size, from the start point a script would hand over (no committed MAP point, no tuning):

    python tools/sample_one_call.py [--seed 2223] [--chains 8] [--iters 300] [--out gpurun_out/sample_one_call.json]   (on an MI355X)

Prints / writes: the stages the recipe went through, acceptance, split-R-hat and multi-chain bulk ESS of the second half."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from nonstationary_multivariate_gaussian_process_amd import drivers, sim  # noqa: E402
import hmc_1000 as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=2223)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--M", type=int, default=3)
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sample_one_call.json"))
    a = ap.parse_args()
    d = sim.simulate_nonseparable(a.N, a.M, seed=a.seed)
    p0 = sim.perturb(d["pars_true"], 0.05, 0.7)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    log = []

    def say(msg):
        log.append("%s %s" % (time.strftime("%H:%M:%S"), msg))
        print(msg, flush=True)
        with open(a.out + ".progress", "a") as f:
            f.write(log[-1] + "\n")
    t0 = time.time()
    S, info = drivers.sample_nonseparable(d["x"], d["Y"], sim.HYPER_SVC, p0, chains=a.chains, iters=a.iters, progress=say)
    dt = time.time() - t0
    Sb = S[a.iters // 2:]
    rh, ess = H.split_rhat(Sb), H.multichain_ess(Sb)
    rec = {"what": "drivers.sample_nonseparable(x, Y, hyper_pars, pars0, chains=%d, iters=%d) on the subject with data seed %d, N = %d, D = %d"
                   % (a.chains, a.iters, a.seed, a.N, a.M),
           "seconds_total": dt, "mode": {k: v for k, v in info["mode"].items() if k != "pars"}, "metric_at_the_mode": info["metric_at_the_mode"],
           "step_search": info.get("step_search"), "step_size": info["step_size"], "stages": info["stages"],
           "second_half": {"draws_per_chain": int(Sb.shape[0]), "split_rhat": H.block_stats(rh), "ess": H.block_stats(ess),
                           "fraction_rhat_below_1.05": float(np.mean(rh < 1.05)), "fraction_rhat_below_1.1": float(np.mean(rh < 1.1))},
           "log": log}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("total %.1f s; R-hat median %.4f max %.4f; ESS median %.0f min %.0f of %d draws" % (
        dt, np.median(rh), rh.max(), np.median(ess), ess.min(), Sb.shape[0] * a.chains))


if __name__ == "__main__":
    main()
