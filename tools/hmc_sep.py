"""The sampler call of the SEPARABLE model (Separable_model.py:209-210: HMC, 20 leapfrog steps, from the MAP estimate) at BASELINE
config 5's size -- N = 4096, D = 5, P = 8,208 -- for `--chains` chains in lock-step under the model's prior-factor metric:

    python tools/hmc_sep.py [--N 4096] [--M 5] [--chains 8] [--iters 300] [--out gpurun_out/hmc_sep.json]       (on an MI355X)

Recipe (drivers.py): mode by metric-preconditioned L-BFGS from the start point (polish_map_separable) -> SeparablePriorMetric at the
mode -> warm-up at a small step -> the metric REBUILT at the chains' mean, twice (the posterior's mass sits far from its mode along
the sigma(x) <-> B scale ridge of this model: log-diagonal of L about 2.2 where the mode has -0.7 at N = 64; the likelihood's
curvature at the typical set is not the mode's) -> step search -> main run.  One batched value+gradient evaluation
(nmgp_sep_batch_eval) per leapfrog step; the two triangular products per step with the prior factors run on the host.
Writes one JSON document: the stages, acceptance, |dH|, samples/s, gradient evaluations/s, multi-chain bulk ESS and split-R-hat per
parameter block (tools/hmc_1000.py's estimators), for the second half of the main run."""
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
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--M", type=int, default=5)
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--iters", type=int, default=250)
    ap.add_argument("--warm", type=int, default=50)
    ap.add_argument("--windows", type=int, default=2)
    ap.add_argument("--window-iters", type=int, default=50)
    ap.add_argument("--leap", type=int, default=20)
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--seed", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hmc_sep.json"))
    ap.add_argument("--progress", default=os.path.join(ROOT, "gpurun_out", "hmc_sep.progress"))
    a = ap.parse_args()
    N, M, B = a.N, a.M, a.chains
    T = M * (M + 1) // 2
    d = sim.simulate_separable(N, M, a.seed)
    h = sim.HYPER_SEP
    os.makedirs(os.path.dirname(a.out), exist_ok=True)

    def note(msg):
        with open(a.progress, "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))
        print(msg, flush=True)

    P = 2 * N + T + 1
    rec = {"config": "BASELINE config 5's model and size: separable GP, D = %d, N = %d (P = %d), one MI355X, %d chains in lock-step, %d "
                     "leapfrog steps per iteration" % (M, N, P, B, a.leap),
           "reference_call": "Separable_model.py:209-210 (step_size 2e-4, num_steps_in_leap 20, identity mass, N_hmc 1000)"}
    p0 = sim.perturb(d["pars_true"], 0.05, 0.4)
    t0 = time.time()
    q0, nl, gn, nev = drivers.polish_map_separable(d["x"], d["Y"], h, p0, maxiter=400, rounds=8, rank=a.rank, probes=a.rank + 32, batch=16,
                                                   verbose=note)
    rec["mode"] = {"log_posterior": -nl, "whitened_gradient_norm": gn, "gradient_evaluations": nev, "seconds": time.time() - t0}
    note("mode: log posterior %.4f, |grad| %.3g, %d evaluations, %.1f s" % (-nl, gn, nev, time.time() - t0))
    t0 = time.time()
    met = drivers.separable_prior_metric(d["x"], d["Y"], h, q0, rank=a.rank, oversample=32, seed=3, batch=16)
    rec["metric_at_the_mode"] = dict({k: v for k, v in met.info.items() if k != "eigenvalues"}, rank=met.rank, seconds=time.time() - t0,
                                     leading_eigenvalues=met.info["eigenvalues"][:24])
    note("metric at the mode: rank %d, lam max %.3g, %.1f s" % (met.rank, met.info["lam_max"], time.time() - t0))
    cur = np.repeat(q0[None], B, 0)
    stages = []

    def run(metric, eps, iters, seed, tag):
        hm = drivers.BatchedHMCSeparable(d["x"], d["Y"], h, cur, step_size=eps, num_steps_in_leap=a.leap, seed=seed, M=metric, step_jitter=0.2)
        t1 = time.time()
        chunks, ees, acc_sum, done = [], [], 0.0, 0
        while done < iters:                      # in segments, so that a long stage keeps reporting
            k = min(25, iters - done)
            s_, info_ = hm.run(k)
            chunks.append(s_)
            ees.append(info_["energy_error"])
            acc_sum += float(info_["accept_rate"].sum()) * k
            done += k
            if iters > 25:
                note("  %s: %d / %d iterations, %.1f s" % (tag, done, iters, time.time() - t1))
        s = np.concatenate(chunks)
        info = {"accept_rate": np.array([acc_sum / (iters * B)]), "energy_error": np.concatenate(ees)}
        dt = time.time() - t1
        # (every segment re-evaluates its start point once: counted)
        st = {"stage": tag, "step_size": eps, "iterations": iters, "seconds": dt, "accept_rate_mean": float(info["accept_rate"].mean()),
              "median_abs_dH": float(np.nanmedian(np.abs(info["energy_error"]))), "grad_evals_per_s": (len(chunks) + iters * a.leap) * B / dt}
        stages.append(st)
        note("%s: %d iterations at eps %.3g in %.1f s, accept %.3f, median |dH| %.3g" % (tag, iters, eps, dt, st["accept_rate_mean"], st["median_abs_dH"]))
        return s, info, dt

    s, _, _ = run(met, 0.04, a.warm, 100, "warm-up (metric at the mode)")
    cur = s[-1]
    for wdw in range(a.windows):
        center = s[-max(10, s.shape[0] // 2):].mean((0, 1))
        t0 = time.time()
        met = drivers.separable_prior_metric(d["x"], d["Y"], h, center, rank=a.rank, oversample=32, seed=5 + wdw, batch=16, factors=met)
        note("window %d: metric at the chains' mean: rank %d, lam max %.3g, most negative %.3g, %.1f s" % (
            wdw, met.rank, met.info["lam_max"], met.info["most_negative"], time.time() - t0))
        s, _, _ = run(met, 0.08, a.window_iters, 200 + wdw, "adaptation window %d (metric at the chains' mean)" % wdw)
        cur = s[-1]
    rec["metric_of_the_main_run"] = dict({k: v for k, v in met.info.items() if k != "eigenvalues"}, rank=met.rank,
                                         leading_eigenvalues=met.info["eigenvalues"][:24])
    best = 0.06
    for eps in (0.08, 0.11, 0.15):
        hm = drivers.BatchedHMCSeparable(d["x"], d["Y"], h, cur, step_size=eps, num_steps_in_leap=a.leap, seed=300, M=met, step_jitter=0.2)
        _, info = hm.run(6)
        acc = float(info["accept_rate"].mean())
        note("step search: eps %.3g accept %.2f median |dH| %.3g" % (eps, acc, np.nanmedian(np.abs(info["energy_error"]))))
        if acc >= 0.8:
            best = eps
        else:
            break
    S, info, dt = run(met, best, a.iters, 1, "main")
    rec["stages"] = stages
    rec["main"] = {"iterations": a.iters, "chains": B, "step_size": best, "seconds": dt, "samples_per_s": a.iters * B / dt,
                   "grad_evals_per_s": (-(-a.iters // 25) + a.iters * a.leap) * B / dt, "accept_rate_mean": float(info["accept_rate"].mean()),
                   "abs_dH": H.block_stats(np.abs(info["energy_error"]))}
    Sb = S[a.iters // 2:]
    blocks = {"tilde_l": np.arange(N), "tilde_sigma": N + np.arange(N), "uL_vec": 2 * N + np.arange(T), "log_sigma2": np.array([P - 1])}
    diag, all_ess, all_rh = {}, [], []
    for name, idx in blocks.items():
        ess, rh = H.multichain_ess(Sb[:, :, idx]), H.split_rhat(Sb[:, :, idx])
        all_ess.append(ess)
        all_rh.append(rh)
        diag[name] = {"ess": H.block_stats(ess), "split_rhat": H.block_stats(rh),
                      "posterior_sd": H.block_stats(Sb[:, :, idx].reshape(-1, idx.size).std(0))}
    all_ess, all_rh = np.concatenate(all_ess), np.concatenate(all_rh)
    rec["diagnostics_second_half"] = {"draws_per_chain": int(Sb.shape[0]), "chains": B, "blocks": diag,
                                      "all_parameters": {"ess": H.block_stats(all_ess), "split_rhat": H.block_stats(all_rh),
                                                         "ess_per_second_median": float(np.median(all_ess) / (dt / 2)),
                                                         "fraction_rhat_below_1.05": float(np.mean(all_rh < 1.05)),
                                                         "fraction_rhat_below_1.2": float(np.mean(all_rh < 1.2))}}
    rec["uL_vec"] = {"mode": q0[2 * N:2 * N + T].tolist(), "posterior_mean": Sb[:, :, 2 * N:2 * N + T].mean((0, 1)).tolist(),
                     "posterior_sd": Sb[:, :, 2 * N:2 * N + T].std((0, 1)).tolist(), "generating": d["pars_true"][2 * N:2 * N + T].tolist()}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    note("wrote %s: %.2f samples/s, accept %.3f, R-hat median %.3f max %.3f, ESS median %.0f" % (
        a.out, rec["main"]["samples_per_s"], rec["main"]["accept_rate_mean"], np.median(all_rh), all_rh.max(), np.median(all_ess)))


if __name__ == "__main__":
    main()
