"""BASELINE config 4's subjects through the whole MCMC recipe on ONE GPU: `--subjects` independent subjects (D = 3, N = 1024, seeds
0..S-1, the mpisim hyper-parameters: Nonseparable_model_mpisim.py:305-348 runs one process per subject), `--chains-per-subject` chains
each, as ONE multi-subject batch under per-subject prior-factor metrics:

    python tools/hmc_subjects.py [--subjects 64] [--chains-per-subject 2] [--iters 300] [--out gpurun_out/hmc_subjects.json]

Per subject: mode by metric-preconditioned L-BFGS from the start point (drivers.polish_map) and its PriorMetric there (rank padded
to the largest: PriorMetric.stack); then all S x k chains in lock-step (drivers.BatchedHMC on nmgp_svc_batch_set_subjects_chains:
every subject with its own data, prior factors and metric directions on the device): a warm-up at a small step, then the main run.
Writes one JSON document: per-stage timings and rates, acceptance, and per SUBJECT the worst split-R-hat and the smallest multi-chain
bulk ESS over its P parameters (tools/hmc_1000.py's estimators on the subject's k chains, second half of the main run)."""
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
    ap.add_argument("--subjects", type=int, default=64)
    ap.add_argument("--chains-per-subject", type=int, default=2)
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--M", type=int, default=3)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--warm", type=int, default=40)
    ap.add_argument("--leap", type=int, default=20)
    ap.add_argument("--step", type=float, default=0.12)
    ap.add_argument("--rank", type=int, default=48)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hmc_subjects.json"))
    ap.add_argument("--progress", default=os.path.join(ROOT, "gpurun_out", "hmc_subjects.progress"))
    a = ap.parse_args()
    S, k, N, M = a.subjects, a.chains_per_subject, a.N, a.M
    T = M * (M + 1) // 2
    P = N * (1 + T) + 1
    h = sim.HYPER_SVC_MPISIM
    os.makedirs(os.path.dirname(a.out), exist_ok=True)

    def note(msg):
        with open(a.progress, "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))
        print(msg, flush=True)

    subs = [sim.simulate_nonseparable(N, M, seed=s) for s in range(S)]
    xs, Ys = np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs])
    rec = {"config": "BASELINE config 4's subjects on one MI355X: %d subjects, D = %d, N = %d (P = %d), %d chains each = %d chains in "
                     "lock-step, %d leapfrog steps per iteration, hyper-parameters of Nonseparable_model_mpisim.py:311-312" % (
                         S, M, N, P, k, S * k, a.leap)}
    t0 = time.time()
    modes, mets, evals = [], [], 0
    for s_, d in enumerate(subs):
        p0 = sim.perturb(d["pars_true"], 0.05, 0.7)
        q, nl, gn, nev = drivers.polish_map(d["x"], d["Y"], h, p0, maxiter=400, rounds=8, rank=a.rank, probes=a.rank + 32)
        met = drivers.prior_lowrank_metric(d["x"], d["Y"], h, q, rank=a.rank, oversample=32, seed=7, batch=a.rank + 32)
        modes.append(q)
        mets.append(met)
        evals += nev + met.info["grad_evals"]
        if s_ % 8 == 7:
            note("modes + metrics: %d / %d subjects, %.1f s (last: log posterior %.3f, |grad| %.3g, rank %d, most negative %.3g)" % (
                s_ + 1, S, time.time() - t0, -nl, gn, met.rank, met.info["most_negative"]))
    rec["modes_and_metrics"] = {"seconds": time.time() - t0, "gradient_evaluations": evals, "ranks": [m.rank for m in mets],
                                "most_negative_eigenvalue_worst": float(min(m.info["most_negative"] for m in mets))}
    metric = drivers.PriorMetric.stack(mets)
    init = np.stack([modes[s_] for s_ in range(S) for _ in range(k)])

    def run(q_start, eps, iters, seed, tag):
        hm = drivers.BatchedHMC(xs, Ys, h, q_start, step_size=eps, num_steps_in_leap=a.leap, seed=seed, M=metric, chains_per_subject=k,
                                step_jitter=0.2)
        t1 = time.time()
        chunks, ees, acc, done = [], [], 0.0, 0
        while done < iters:
            n_ = min(25, iters - done)
            s_, info = hm.run(n_)
            chunks.append(s_)
            ees.append(info["energy_error"])
            acc += float(info["accept_rate"].sum()) * n_
            done += n_
            note("  %s: %d / %d iterations, %.1f s" % (tag, done, iters, time.time() - t1))
        dt = time.time() - t1
        st = {"stage": tag, "step_size": eps, "iterations": iters, "seconds": dt, "accept_rate_mean": acc / (iters * S * k),
              "median_abs_dH": float(np.nanmedian(np.abs(np.concatenate(ees)))), "grad_evals_per_s": (len(chunks) + iters * a.leap) * S * k / dt,
              "samples_per_s": iters * S * k / dt}
        note("%s: accept %.3f, median |dH| %.3g, %.0f gradient evals/s" % (tag, st["accept_rate_mean"], st["median_abs_dH"], st["grad_evals_per_s"]))
        return np.concatenate(chunks), st

    sw, st_w = run(init, 0.03, a.warm, 100, "warm-up")
    Sm, st_m = run(sw[-1], a.step, a.iters, 1, "main")
    rec["stages"] = [st_w, st_m]
    Sb = Sm[a.iters // 2:]
    per_subject = []
    for s_ in range(S):
        ch = Sb[:, s_ * k:(s_ + 1) * k]
        rh, ess = H.split_rhat(ch), H.multichain_ess(ch)
        per_subject.append({"subject": s_, "split_rhat_median": float(np.nanmedian(rh)), "split_rhat_max": float(np.nanmax(rh)),
                            "ess_median": float(np.median(ess)), "ess_min": float(ess.min())})
    rmax = np.array([p["split_rhat_max"] for p in per_subject])
    emin = np.array([p["ess_min"] for p in per_subject])
    rec["diagnostics_second_half"] = {"draws_per_chain": int(Sb.shape[0]), "chains_per_subject": k,
                                      "worst_split_rhat_over_subjects": H.block_stats(rmax), "smallest_ess_over_subjects": H.block_stats(emin),
                                      "subjects_with_all_rhat_below_1.1": int(np.sum(rmax < 1.1)), "per_subject": per_subject}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    note("wrote %s: %.1f samples/s, %.0f gradient evals/s, accept %.3f; worst R-hat per subject: median %.3f max %.3f; %d / %d subjects below 1.1" % (
        a.out, st_m["samples_per_s"], st_m["grad_evals_per_s"], st_m["accept_rate_mean"], np.median(rmax), rmax.max(), int(np.sum(rmax < 1.1)), S))


if __name__ == "__main__":
    main()
