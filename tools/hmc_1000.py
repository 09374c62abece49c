"""BASELINE config 3 as worded: the nonseparable GP at N = 2048, D = 3 on one MI355X, 1000 MCMC iterations.

    python tools/hmc_1000.py [--chains 8] [--iters 1000] [--mass prior|diag|identity] [--rank 96] [--pilot 100]
                             [--step auto|<eps>] [--out gpurun_out/hmc_1000.json] [--progress gpurun_out/hmc_1000.progress]

The sampler call of Nonseparable_model.py:228-231 (HMC, 20 leapfrog steps per iteration, started from the MAP estimate,
duplicate_samples=True) run for `--chains` chains of the bench subject in lock-step (drivers.BatchedHMC: trajectories resident in
HBM, one batched value+gradient launch sequence per leapfrog step).  The reference's production runs pass a mass matrix derived
from a previous run (Nonseparable_model_mpiKAISER.py:398-411).  `--mass prior` (default): the prior-factor metric
M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T (drivers.PriorMetric; csrc/nmgp_metric.hip) -- the cached GP-prior Cholesky factors
plus a rank-`--rank` correction for the likelihood's curvature found by randomised Hessian-vector products at the MAP point
(drivers.prior_lowrank_metric, a few seconds); `--mass diag`: `--pilot` identity-mass iterations give a per-parameter scale,
M = diag(1 / var) (round 4: does not mix).  The step size is chosen by a short search for ~0.8 acceptance (`--step auto`; the
reference's 1e-4 is rejected every time at this size under the identity: profiles/r03_hmc_steps.txt).

(`drivers.sample_nonseparable` packages the same recipe -- mode, metric, warm-up, step search, run -- behind one call; this tool keeps
its own spelling because it also runs the round-4 comparisons `--mass diag|identity` and writes the diagnostics.)
Writes one JSON document: acceptance, quantiles of |dH|, samples/s and gradient evaluations/s of the main run, the MULTI-CHAIN
effective sample size (rank-normalised split chains, between-chain variance in the denominator: Vehtari et al. 2021 -- chains that
have not mixed get a small ESS, unlike a sum of per-chain figures), ESS per second, and split-R-hat per parameter block (l~(x), the T columns of uL(x), log sigma^2), the posterior mean of l~(x) against the generating curve and the MAP estimate,
and the rms distance the chains travelled."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonstationary_multivariate_gaussian_process_amd import drivers, sim  # noqa: E402


def autocorr_ess(x):
    """x [S, C, K]: S draws of C chains for K parameters -> ESS [K] (sum over chains of the per-chain ESS)."""
    S = x.shape[0]
    xc = x - x.mean(0, keepdims=True)
    nfft = 1 << (2 * S - 1).bit_length()
    f = np.fft.rfft(xc, n=nfft, axis=0)
    ac = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:S]
    var0 = ac[0].copy()
    var0[var0 <= 0] = np.inf
    rho = ac / var0
    # Geyer: sum of adjacent pairs while positive
    pairs = rho[0:S - (S % 2):2] + rho[1:S - (S % 2) + 1:2]
    pos = np.cumprod(pairs > 0, axis=0).astype(bool)
    tau = -1.0 + 2.0 * np.where(pos, pairs, 0.0).sum(0)
    tau = np.maximum(tau, 1.0 / S)
    ess = S / tau
    return np.minimum(ess, S).sum(0)


def split_rhat(x):
    """x [S, C, K] -> split-R-hat [K]."""
    S = x.shape[0] // 2 * 2
    h = S // 2
    y = np.concatenate([x[:h], x[h:S]], axis=1)            # [h, 2C, K]
    m = y.mean(0)
    W = y.var(0, ddof=1).mean(0)
    Bv = h * m.var(0, ddof=1)
    W = np.where(W <= 0, np.nan, W)
    return np.sqrt(((h - 1) / h * W + Bv / h) / W)


def multichain_ess(x, rank_normalize=True):
    """x [S, C, K] -> bulk ESS [K] of the C chains TOGETHER (Stan's estimator: split chains, rank-normalised draws, autocorrelation
    rho_t = 1 - (W - mean_c acov_t) / var_plus with the between-chain variance inside var_plus, Geyer's initial monotone positive
    sequence).  Chains stuck at different places give var_plus >> W, rho_t ~ 1 and an ESS of a handful."""
    from scipy.stats import norm, rankdata
    S = x.shape[0] // 2 * 2
    n = S // 2
    y = np.concatenate([x[:n], x[n:S]], axis=1)             # [n, m, K]
    m, K = y.shape[1], y.shape[2]
    if rank_normalize:
        r = rankdata(y.reshape(n * m, K), axis=0)
        y = norm.ppf((r - 0.375) / (n * m + 0.25)).reshape(n, m, K)
    yc = y - y.mean(0, keepdims=True)
    nfft = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(yc, n=nfft, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:n] / n          # biased autocovariance per chain
    chain_var = acov[0] * n / (n - 1.0)
    W = chain_var.mean(0)
    var_plus = W * (n - 1.0) / n
    if m > 1:
        var_plus = var_plus + y.mean(0).var(0, ddof=1)
    var_plus = np.where(var_plus <= 0, np.inf, var_plus)
    rho = 1.0 - (W[None] - acov.mean(1)) / var_plus[None]                # [n, K]
    rho[0] = 1.0
    npair = n // 2
    pairs = rho[0:2 * npair:2] + rho[1:2 * npair:2]                      # [npair, K]
    pos = np.cumprod(pairs > 0, axis=0).astype(bool)
    pairs = np.where(pos, pairs, 0.0)
    pairs = np.minimum.accumulate(pairs, axis=0)                         # initial monotone sequence
    tau = -1.0 + 2.0 * pairs.sum(0)
    tau = np.maximum(tau, 1.0 / np.log10(max(n * m, 10)))
    return n * m / tau


def block_stats(v):
    v = np.asarray(v, dtype=np.float64)
    v = v[np.isfinite(v)]
    q = np.quantile(v, [0.0, 0.05, 0.5, 0.95, 1.0]) if v.size else [np.nan] * 5
    return {"min": float(q[0]), "q05": float(q[1]), "median": float(q[2]), "q95": float(q[3]), "max": float(q[4]),
            "mean": float(v.mean()) if v.size else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--pilot", type=int, default=100)
    ap.add_argument("--mass", choices=["prior", "diag", "identity"], default="prior")
    ap.add_argument("--rank", type=int, default=96)
    ap.add_argument("--probe-h", type=float, default=1e-3)
    ap.add_argument("--warm", type=int, default=40, help="mass prior: thermalising iterations from the MAP point at a small step before the step search")
    ap.add_argument("--jitter", type=float, default=0.2, help="mass prior: the step of iteration i is eps (1 + jitter u_i), u_i ~ U(-1, 1)")
    ap.add_argument("--save-state", default="", help="write the polished MAP point and the chains' final (typical-set) positions to this .npz "
                                                     "(bench.py's `hmc` object starts from it: tests/golden/hmc_state_N2048_M3_seed2222.npz)")
    ap.add_argument("--polish", type=int, default=300, help="L-BFGS iterations on the committed Adam MAP estimate before the metric is built (0 = none)")
    ap.add_argument("--step", default="auto")
    ap.add_argument("--leap", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hmc_1000.json"))
    ap.add_argument("--progress", default=os.path.join(ROOT, "gpurun_out", "hmc_1000.progress"))
    a = ap.parse_args()
    N, M, seed = 2048, 3, 2222
    T = M * (M + 1) // 2
    B = a.chains
    d = sim.simulate_nonseparable(N, M, seed=seed)
    g = np.load(os.path.join(ROOT, "tests", "golden", "map_N%d_M%d_seed%d.npz" % (N, M, seed)))
    qmap = g["pars_map"]
    P = qmap.shape[0]
    q0 = np.repeat(qmap[None], B, 0)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)

    def note(msg):
        with open(a.progress, "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))
        print(msg, flush=True)

    rec = {"config": "BASELINE config 3: nonseparable GP, D = 3, N = 2048 (P = %d), one MI355X, %d HMC iterations x %d chains in "
                     "lock-step, %d leapfrog steps per iteration, start = MAP estimate (tests/golden/%s)" % (
                         P, a.iters, B, a.leap, "map_N%d_M%d_seed%d.npz" % (N, M, seed)),
           "reference_call": "Nonseparable_model.py:228-231 (step_size 1e-4, num_steps_in_leap 20, identity mass)"}
    eps_id = 4e-5
    mass_kw = {}
    if a.mass == "prior":
        if a.polish > 0:
            t0 = time.time()
            qp, nl, gn, nev = drivers.polish_map(d["x"], d["Y"], sim.HYPER_SVC, qmap, maxiter=a.polish, verbose=note)
            rec["map_polish"] = {"lbfgs_iterations_max": a.polish, "evaluations": nev, "seconds": time.time() - t0,
                                 "log_posterior_before": float(g["target_value_hist"][-1]), "log_posterior_after": -nl,
                                 "whitened_gradient_norm_after": gn, "rms_change_per_parameter": float(np.sqrt(np.mean((qp - qmap) ** 2)))}
            note("polish: %d evaluations in %.1f s, log posterior %.4f -> %.4f, |grad| %.3g" % (
                nev, time.time() - t0, float(g["target_value_hist"][-1]), -nl, gn))
            qmap = qp
            q0 = np.repeat(qmap[None], B, 0)
        t0 = time.time()
        met = drivers.prior_lowrank_metric(d["x"], d["Y"], sim.HYPER_SVC, qmap, rank=a.rank, oversample=32, power_iters=1, h=a.probe_h,
                                           seed=7, batch=128)
        dtm = time.time() - t0
        rec["metric"] = dict(met.info, seconds=dtm, rank=met.rank)
        mass_kw = {"M": met}
        q_start = q0
        if a.warm > 0:
            # All chains start AT the mode: potential minimal, so the first trajectories turn P/2 ~ 7000 units of kinetic energy into
            # potential energy and the leapfrog error of that transfer (~ eps^2 E / 8) would reject every step size worth having.
            # A few iterations at a small step put the chains into the typical set; the step search starts from there.
            t0 = time.time()
            hw = drivers.BatchedHMC(d["x"], d["Y"], sim.HYPER_SVC, q0, step_size=0.03, num_steps_in_leap=a.leap, seed=300, **mass_kw)
            sw, iw = hw.run(a.warm)
            q_start = sw[-1]
            rec["warm_up"] = {"iterations": a.warm, "step_size": 0.03, "seconds": time.time() - t0,
                              "accept_rate_mean": float(iw["accept_rate"].mean()),
                              "median_abs_dH_first_5": float(np.nanmedian(np.abs(iw["energy_error"][:5]))),
                              "median_abs_dH_last_10": float(np.nanmedian(np.abs(iw["energy_error"][-10:])))}
            note("warm-up: %d iterations at eps 0.03 in %.1f s, accept %.2f, median |dH| first 5 %.3g, last 10 %.3g" % (
                a.warm, time.time() - t0, iw["accept_rate"].mean(), rec["warm_up"]["median_abs_dH_first_5"],
                rec["warm_up"]["median_abs_dH_last_10"]))
        note("metric: rank %d of %d probes in %.1f s (%d gradient evaluations), lam max %.3g, last kept %.3g, first dropped %s" % (
            met.rank, met.info["probes"], dtm, met.info["grad_evals"], met.info["lam_max"], met.info["lam_min_kept"],
            met.info["first_dropped"]))
    elif a.mass == "diag":
        t0 = time.time()
        hp = drivers.BatchedHMC(d["x"], d["Y"], sim.HYPER_SVC, q0, step_size=eps_id, num_steps_in_leap=a.leap, seed=100)
        sp, ip = hp.run(a.pilot)
        dtp = time.time() - t0
        # per-parameter scale: mean squared displacement from the start over the second half of the pilot, all chains
        half = sp[a.pilot // 2:]
        var = ((half - qmap[None, None]) ** 2).mean((0, 1))
        # regularise: the pilot barely moves the slow directions -- keep the ratio of scales within 1e4 and smooth within blocks
        var = np.clip(var, var.max() * 1e-8, None)
        rec["pilot"] = {"iterations": a.pilot, "step_size": eps_id, "seconds": dtp, "accept_rate_mean": float(ip["accept_rate"].mean()),
                        "scale_quantiles": block_stats(np.sqrt(var))}
        q_start = sp[-1]                                   # continue from where the pilot ended
        mass_kw = {"M": 1.0 / var, "Minv": var}
        note("pilot: %d iterations in %.1f s, accept %.2f, scale median %.2e" % (a.pilot, dtp, ip["accept_rate"].mean(), np.sqrt(np.median(var))))
    else:
        q_start = q0
    # step size: short search for ~0.8 acceptance
    if a.step == "auto":
        # diag mass: the drift per step is eps * scale_i * z_i, so eps is in units of the pilot's scales; start where the median
        # parameter moves as far per step as under the identity mass at eps_id
        if a.mass == "prior":
            # whitened coordinates: the posterior is ~N(0, I) in P dimensions, so eps ~ P^-1/4; 20 steps of 0.08 = a quarter period
            cands = [0.05, 0.065, 0.08, 0.1, 0.12]
        else:
            base = eps_id if a.mass == "identity" else eps_id / float(np.sqrt(np.median(mass_kw["Minv"])))
            cands = [base * f for f in ((0.5, 1.0, 2.0) if a.mass == "identity" else (0.5, 1.0, 2.0, 4.0, 8.0, 16.0))]
        tried = []
        best = None
        for eps in cands:
            hs = drivers.BatchedHMC(d["x"], d["Y"], sim.HYPER_SVC, q_start, step_size=eps, num_steps_in_leap=a.leap, seed=200, **mass_kw)
            _, isr = hs.run(4 if a.mass != "prior" else 6)
            acc = float(isr["accept_rate"].mean())
            med = float(np.nanmedian(np.abs(isr["energy_error"])))
            tried.append({"step_size": eps, "accept_rate_mean": acc, "median_abs_dH": med})
            note("step search: eps %.3g accept %.2f median |dH| %.3g" % (eps, acc, med))
            if acc >= 0.7:
                best = eps
            elif best is not None:
                break
        eps = best if best is not None else cands[0]
        rec["step_search"] = tried
    else:
        eps = float(a.step)
    rec["step_size"] = eps
    rec["mass"] = {"diag": "diagonal, M = 1 / (mean squared displacement of the pilot), momenta and energies on the device (nmgp_svc_batch_traj_z)",
                   "identity": "identity",
                   "prior": "prior-factor metric M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T (nmgp_svc_batch_traj_set_mass_prior): cached "
                            "GP-prior Cholesky factors + rank-%d likelihood correction, whitened momenta on the device" % (
                                mass_kw["M"].rank if a.mass == "prior" else 0)}[a.mass]
    jit = a.jitter if a.mass == "prior" else 0.0
    rec["step_jitter"] = jit
    hm = drivers.BatchedHMC(d["x"], d["Y"], sim.HYPER_SVC, q_start, step_size=eps, num_steps_in_leap=a.leap, seed=1, step_jitter=jit,
                            **mass_kw)
    # run in segments so that progress is visible
    seg = 50
    chunks, accs, ees = [], [], []
    t0 = time.time()
    done = 0
    t_traj = t_loop = 0.0
    while done < a.iters:
        k = min(seg, a.iters - done)
        s, info = hm.run(k)
        t_traj += info["timing"]["trajectory_call_seconds"]
        t_loop += info["timing"]["loop_seconds"]
        chunks.append(s)
        accs.append(info["accept_rate"] * k)
        ees.append(info["energy_error"])
        done += k
        note("main: %d / %d iterations, %.1f s, accept so far %.3f" % (done, a.iters, time.time() - t0, np.sum(accs) / (done * B)))
    dt = time.time() - t0
    S = np.concatenate(chunks)                              # [iters, B, P]
    ee = np.concatenate(ees)
    acc = np.sum(accs, 0) / a.iters
    # (every hm.run() call re-evaluates the start point once: iters / seg extra evaluations, counted)
    evals = (a.iters * a.leap + len(chunks)) * B
    rec["main"] = {"iterations": a.iters, "chains": B, "seconds": dt, "samples_per_s": a.iters * B / dt,
                   "grad_evals_per_s": evals / dt, "accept_rate_by_chain": acc.tolist(), "accept_rate_mean": float(acc.mean()),
                   "abs_dH": block_stats(np.abs(ee)), "dH_mean": float(np.nanmean(ee)),
                   "device_share_of_the_sampling_loop": t_traj / max(t_loop, 1e-12)}
    # diagnostics on the second half (the first half as burn-in)
    Sb = S[a.iters // 2:]
    blocks = {"tilde_l": np.arange(N)}
    for t in range(T):
        blocks["uL_col%d" % t] = N + np.arange(N) * T + t
    blocks["log_sigma2"] = np.array([P - 1])
    diag = {}
    all_ess, all_rh = [], []
    for name, idx in blocks.items():
        ess = multichain_ess(Sb[:, :, idx])
        rh = split_rhat(Sb[:, :, idx])
        all_ess.append(ess)
        all_rh.append(rh)
        diag[name] = {"ess": block_stats(ess), "ess_per_chain_summed_r04_definition": block_stats(autocorr_ess(Sb[:, :, idx])),
                      "split_rhat": block_stats(rh), "posterior_sd": block_stats(Sb[:, :, idx].reshape(-1, idx.size).std(0))}
    all_ess, all_rh = np.concatenate(all_ess), np.concatenate(all_rh)
    # the second half was produced in half the main run's time
    rec["diagnostics_second_half"] = {
        "draws_per_chain": int(Sb.shape[0]), "chains": B, "blocks": diag,
        "all_parameters": {"ess": block_stats(all_ess), "split_rhat": block_stats(all_rh),
                           "ess_per_second_median": float(np.median(all_ess) / (dt / 2)), "ess_per_second_min": float(all_ess.min() / (dt / 2)),
                           "fraction_rhat_below_1.05": float(np.mean(all_rh < 1.05)), "fraction_rhat_below_1.2": float(np.mean(all_rh < 1.2))},
        "note": "ESS = multi-chain bulk ESS (rank-normalised split chains, between-chain variance in var_plus; Vehtari et al. 2021): at "
                "most ~draws x chains; chains that have not mixed get a small value.  The round-4 figure (sum of per-chain ESS) is kept "
                "beside it for comparison only"}
    tl_true = d["pars_true"][:N]
    tl_mean = Sb[:, :, :N].mean((0, 1))
    rec["tilde_l_curve"] = {"rms_posterior_mean_minus_truth": float(np.sqrt(np.mean((tl_mean - tl_true) ** 2))),
                            "rms_map_minus_truth": float(np.sqrt(np.mean((qmap[:N] - tl_true) ** 2))),
                            "rms_posterior_mean_minus_map": float(np.sqrt(np.mean((tl_mean - qmap[:N]) ** 2))),
                            "at_x": [float(v) for v in d["x"][::256]], "truth": [float(v) for v in tl_true[::256]],
                            "posterior_mean": [float(v) for v in tl_mean[::256]], "map": [float(v) for v in qmap[:N][::256]],
                            "posterior_sd": [float(v) for v in Sb[:, :, :N].reshape(-1, N).std(0)[::256]]}
    rec["rms_displacement_from_start_per_parameter"] = float(np.sqrt(np.mean((S[-1] - q_start) ** 2)))
    rec["log_sigma2"] = {"truth": float(d["pars_true"][-1]), "map": float(qmap[-1]), "posterior_mean": float(Sb[:, :, -1].mean()),
                         "posterior_sd": float(Sb[:, :, -1].std())}
    if a.save_state:
        np.savez_compressed(a.save_state, pars_polished=qmap, pars_typical=S[-1], step_size=eps, iterations=a.iters, chains=B,
                            accept_rate_mean=float(acc.mean()), seed=seed, N=N, M=M,
                            made_by="tools/hmc_1000.py --mass %s --iters %d --chains %d (GPU run; see profiles/r05_hmc_1000.json)" % (a.mass, a.iters, B))
        note("state written to %s" % a.save_state)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    note("wrote %s: %.2f samples/s, accept %.3f" % (a.out, rec["main"]["samples_per_s"], acc.mean()))


if __name__ == "__main__":
    main()
