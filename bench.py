#!/usr/bin/env python
"""Headline benchmark: log-posterior evaluations / second of the nonseparable GP (D=3 outputs, N=2048 locations,
MN = 6144) on MI355X -- BASELINE.json's metric on its configs[2].

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input: ONE evaluation of ``nlogpos_obj_SVC`` for each
of ``--chains`` (default 128) independent MCMC chains of the rank's subject -- the chains are the reference's
embarrassingly-parallel unit (it runs them as separate processes, Nonseparable_model_mpisim.py:305-306); here their
parameter vectors are stacked [B, P] in HBM and one launch sequence evaluates all of them (nmgp_svc_batch_*), which is
what amortises the latency-bound panel steps of the Cholesky.  ``value`` counts evaluations: steps x chains x GPUs /
time.  ``--chains 1`` gives the single-chain latency path (``--grad`` adds the gradient there).  Data follow the
reference simulator's recipe (SIM_code/sim.py:177-263); the subject's data and all parameter vectors are resident in
HBM when the timed region starts, and the host reads back the verbose scalars of every chain after every step as an
MCMC driver would.  With N GPUs every rank owns its own subject and chains: weak scaling, no collective on the data
path; RCCL is used for the barrier, the max-over-ranks time and the final reduction of the chains' statistics only.

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline      -- the dominant kernel (the FP64 Cholesky factorisation of the 6144^2 covariance): algorithmic
                   n^3/3 flop divided by its average duration measured with HIP events on the library's stream.
  cpu_baseline  -- the NumPy/SciPy oracle (oracle/nmgp_oracle.py, Cholesky formulation) timed on this host's cores
                   on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix (vendor dense figure).  The CDNA4 guide lists no
                                   # FP64 row; the measured rocBLAS dgemm rate is reported beside it in `config`.
HBM_PEAK_GBS = 8000.0
# HBM bytes per batched factorisation from the rocprofv3 PMC passes (profiles/, filled in by hand from the committed
# counter CSVs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None until measured.
TRAFFIC_BYTES_PER_LAUNCH = {(2048, 3, 128): 3.907e11, (2048, 3, 64): 1.800e11}    # (N, M, chains) -> HBM bytes of the k_syrk_lower launches of one
#                                                         batched factorisation; profiles/r01_v9_batched128_pmc_traffic.json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--M", type=int, default=3)
    ap.add_argument("--grad", action="store_true", help="time value+gradient evaluations")
    ap.add_argument("--workload", choices=["chain", "subjects"], default="chain",
                    help="chain: one N=2048 chain per GPU (headline); subjects: BASELINE config 4, independent "
                         "subjects of size --N sharded round-robin over the GPUs (8 per GPU), one stream each")
    ap.add_argument("--subjects-per-gpu", type=int, default=8)
    ap.add_argument("--chains", type=int, default=128,
                    help="independent MCMC chains of the subject evaluated per step through the batched entry "
                         "(nmgp_svc_batch_*): one launch sequence covers all chains")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the chains into this many groups, each a batched context on its own pair of HIP streams, "
                         "so that one group's latency-bound panel steps overlap another group's MFMA updates")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-evals", type=int, default=3)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed launch (WORLD_SIZE=%d)" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from nonstationary_multivariate_gaussian_process_amd import _lib, chains, sim

    if a.workload == "subjects":
        return bench_subjects(a, rank, world, local_rank, torch, dist, _lib, chains, sim)
    N, M = a.N, a.M
    n = N * M
    # one independent subject per rank: seed 2222 is the reference's single-subject seed (sim.py:359)
    d = sim.simulate_nonseparable(N, M, seed=2222 + rank)
    pars = sim.perturb(d["pars_true"], 0.05, 0.7)
    hyper = sim.HYPER_SVC
    hv = np.array([hyper[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a",
                                      "b")], dtype=np.float64)
    ctx = _lib.Context(local_rank)
    ctx.set_data(d["x"], d["Y"])
    ctx.svc_set_pars(pars)
    want_grad = bool(a.grad)
    B = max(1, a.chains)
    G = max(1, min(a.groups, B))
    ctxs = [ctx] + [_lib.Context(local_rank) for _ in range(G - 1)]
    if B > 1:
        # chain b starts from its own smooth perturbation of the generating parameters
        allp = np.stack([sim.perturb(d["pars_true"], 0.05, 0.7 + 0.37 * b) for b in range(B)])
        sizes = [B // G + (1 if g < B % G else 0) for g in range(G)]
        off = 0
        for cg, sz in zip(ctxs, sizes):
            cg.set_data(d["x"], d["Y"])
            cg.svc_batch_alloc(sz)
            cg.svc_batch_set_pars(allp[off:off + sz])
            off += sz

    def step():
        if B > 1:
            for cg in ctxs:
                cg.svc_batch_eval(hv, True, want_grad)
            first = None
            for cg in ctxs:
                o, st = cg.svc_batch_fetch()
                if st.any():
                    raise RuntimeError("chain failed: %s" % st)
                first = o[0] if first is None else first
            return first
        ctx.svc_eval_resident(hv, True, want_grad)
        return ctx.svc_fetch(False)[0]

    # stage timers on from the warm-up on: their HIP events are created on first use, which must not fall into the timed region
    ctx.profile_enable(True)
    for _ in range(a.warmup):
        out = step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for cg in ctxs:
            cg.sync()

    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read()
    # dominant-kernel pass: the same K steps again with one HIP-event pair around EVERY k_syrk_lower launch (on the
    # stream it is launched on); kept out of the timed region so the ~200 extra event records per factorisation do not
    # leak into `value`
    ctx.profile_enable(2)
    ctx.profile_reset()
    for _ in range(a.steps):
        step()
    barrier()
    kprof = ctx.profile_read_work()
    ctx.profile_enable(False)

    elapsed_max = max_over_ranks(elapsed, world, torch, dist)
    # the reduction step of the per-subject chains (RCCL all-reduce + all-gather of 8-double rows)
    row = np.array([[rank, 1.0, a.steps] + [float(v) for v in out[:5]]])
    chain_stats, chain_table = chains.reduce_rows(row, world, world, device="cuda")
    total_evals = a.steps * world * B
    value = total_evals / elapsed_max

    if rank == 0:
        chol_ms, chol_cnt = prof["chol"]
        chol_avg_s = (chol_ms / max(chol_cnt, 1)) * 1e-3
        B0 = (B // G + (1 if B % G else 0)) if B > 1 else 1     # chains in the profiled context (group 0)
        flops = B0 * n ** 3 / 3.0
        chol_tf = flops / chol_avg_s / 1e12 if chol_avg_s > 0 else 0.0
        # k_syrk_lower: sum of algorithmic flop (2K per updated lower-trapezoid element) / sum of launch durations
        syrk_ms, syrk_cnt, syrk_flop, syrk_bytes = kprof["syrk"]
        achieved = syrk_flop / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        syrk_info = {"launches_per_step": syrk_cnt / max(a.steps, 1), "avg_launch_us": 1e3 * syrk_ms / max(syrk_cnt, 1),
                     "ms_per_step": syrk_ms / max(a.steps, 1), "gflop_per_step": syrk_flop / max(a.steps, 1) / 1e9,
                     "algorithmic_bytes_per_step": syrk_bytes / max(a.steps, 1),
                     "share_of_factorisation_flop": syrk_flop / max(a.steps, 1) / flops}
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in prof.items() if v[1] > 0}
        cov_ms = stage_ms.get("cov", 0.0)
        cov_bytes = B0 * 8.0 * n * (n + 1) / 2.0
        try:
            dgemm_tf = ctx.measure_dgemm_tflops(4096, 5)
            hbm_gbs = ctx.measure_hbm_gbs(1 << 30, 10)
        except Exception:       # measurement helpers are informative only
            dgemm_tf, hbm_gbs = None, None
        rec = {
            "metric": "log-posterior evals/sec (N=2048, D=3 nonseparable GP)" if (N, M) == (2048, 3) else
                      "log-posterior evals/sec (N=%d, D=%d nonseparable GP)" % (N, M),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed_max / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "nonseparable GP nlogpos_obj_SVC %s, D=%d, N=%d (MN=%d), %d independent chain(s) per "
                                   "GPU evaluated per step" % ("value+gradient" if want_grad else "value", M, N, n, B),
                       "chains_per_gpu": B, "stream_groups": G,
                       "stage_ms": stage_ms, "neglog_rank0": float(out[0]),
                       "chains_ok": int(chain_stats[0]), "sum_neglog_all_chains": float(chain_stats[3]),
                       "measured_dgemm_tflops_n4096": dgemm_tf, "measured_hbm_copy_gbs": hbm_gbs,
                       "cov_build_gbs": (cov_bytes / (cov_ms * 1e-3) / 1e9) if cov_ms > 0 else None},
            "roofline": {"kernel": "k_syrk_lower (v_mfma_f64_16x16x4_f64 trailing update of the blocked FP64 Cholesky of "
                                   "%d %dx%d covariances): achieved = sum over launches of 2K*(updated lower-trapezoid "
                                   "elements) / sum of HIP-event launch durations on the launching stream" % (B, n, n),
                         "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH.get((N, M, B)),
                         "traffic_note": "HBM bytes of the k_syrk_lower launches of ONE step (same scope as syrk.gflop_per_step), from "
                                         "the rocprofv3 FETCH_SIZE (x2, gfx950) / WRITE_SIZE passes committed under profiles/",
                         "syrk": syrk_info,
                         "factorisation": {"what": "whole CHOL stage (syrk + potf2 + trsm + row moves), chains*n^3/3 flop "
                                                   "over the HIP-event stage time of the timed region",
                                           "ms": 1e3 * chol_avg_s, "achieved": chol_tf,
                                           "frac": chol_tf / FP64_MATRIX_PEAK_TFLOPS}},
        }
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(d, pars, hyper, a.cpu_evals, want_grad)
        print(json.dumps(rec), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def max_over_ranks(elapsed, world, torch, dist):
    if world == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def bench_subjects(a, rank, world, local_rank, torch, dist, _lib, chains, sim):
    """BASELINE config 4 (Nonseparable_model_mpisim-style): subjects_per_gpu x world independent subjects, D=M, N=--N
    (default there: 1024), subject s on rank s mod world (the reference maps rank -> data file).  The rank's subjects
    form ONE multi-subject batch (own x, Y and prior factors per batch element, nmgp_svc_batch_set_subjects): a step =
    one evaluation of EVERY local subject by one launch sequence."""
    N, M = a.N, a.M
    n_subj = a.subjects_per_gpu * world
    mine = chains.partition(n_subj, world, rank)
    hyper = sim.HYPER_SVC_MPISIM
    hv = np.array([hyper[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a",
                                      "b")], dtype=np.float64)
    subs = [sim.simulate_nonseparable(N, M, seed=s_id) for s_id in mine]       # subject s uses seed s (sim.py:361-363)
    ctx = _lib.Context(local_rank)
    ctx.set_data(subs[0]["x"], subs[0]["Y"])
    ctx.svc_batch_alloc(len(mine))
    ctx.svc_batch_set_subjects(np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs]))
    ctx.svc_batch_set_pars(np.stack([sim.perturb(d["pars_true"], 0.05, 0.7) for d in subs]))
    want_grad = bool(a.grad)

    def step():
        ctx.svc_batch_eval(hv, True, want_grad)
        return ctx.svc_batch_fetch()

    ctx.profile_enable(True)        # stage timers (their events are created during the warm-up)
    for _ in range(a.warmup):
        outs, status = step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs, status = step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, torch, dist)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in prof.items() if v[1] > 0}
    n = N * M
    chol_s = stage_ms.get("chol", 0.0) * 1e-3
    fact_tf = (len(mine) * (2.0 if want_grad else 1.0) * n ** 3 / 3.0) / chol_s / 1e12 if chol_s > 0 else 0.0
    rows = np.array([[s_id, float(st == 0), a.steps] + [float(v) for v in o[:5]]
                     for s_id, o, st in zip(mine, outs, status)])
    stats, table = chains.reduce_rows(rows, n_subj, world, device="cuda")
    if rank == 0:
        total = a.steps * n_subj
        print(json.dumps({
            "metric": "log-posterior evals/sec (%d subjects, N=%d, D=%d nonseparable GP)" % (n_subj, N, M),
            "value": total / elapsed, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d independent subjects (%d per GPU, one multi-subject batch per GPU), "
                                   "nlogpos_obj_SVC %s, D=%d, N=%d" % (n_subj, a.subjects_per_gpu,
                                                                       "value+gradient" if want_grad else "value", M, N),
                       "subjects_ok": int(stats[0]), "sum_neglog": float(stats[3]), "stage_ms": stage_ms},
            "roofline": {"kernel": "blocked FP64 Cholesky stage of rank 0's multi-subject batch (k_syrk_lower + panel kernels%s): "
                                   "subjects * n^3/3 flop%s over the HIP-event stage time" % (
                                       ", with the L^-T rows" if want_grad else "", " x 2" if want_grad else ""),
                         "bound": "mfma", "achieved": fact_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fact_tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None}}), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(d, pars, hyper, evals, want_grad):
    """The CPU oracle on the same subject: 1 warm-up + `evals` timed evaluations (Cholesky formulation); `cores` = the
    BLAS/LAPACK threads NumPy/SciPy actually use on this host (threadpoolctl), else the affinity mask."""
    from oracle import nmgp_oracle as O     # checker / baseline only
    cores = len(os.sched_getaffinity(0))
    try:        # the threads the BLAS/LAPACK behind NumPy/SciPy actually runs (the elementwise parts are single-threaded)
        from threadpoolctl import threadpool_info
        blas = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = int(max(blas))
    except Exception:
        pass

    def one():
        return O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **hyper, verbose=True, formulation="cholesky", grad=want_grad)
    one()
    ts = []
    for _ in range(evals):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": 1.0 / med, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d evaluations of the same N=%d, D=%d subject (median %.2f s each), NumPy/SciPy oracle, "
                      "Cholesky formulation, %s" % (evals, d["Y"].shape[0], d["Y"].shape[1], med,
                                                    "value+gradient" if want_grad else "value only")}


if __name__ == "__main__":
    main()
