#!/usr/bin/env python
"""Headline benchmark: log-posterior evaluations / second of the nonseparable GP (D=3 outputs, N=2048 locations,
MN = 6144) on MI355X -- BASELINE.json's metric on its configs[2].

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input: ONE evaluation of ``nlogpos_obj_SVC`` for each
of ``--chains`` (default 128) independent MCMC chains of the rank's subject -- the chains are the reference's
embarrassingly-parallel unit (it runs them as separate processes, Nonseparable_model_mpisim.py:305-306); here their
parameter vectors are stacked [B, P] in HBM and one launch sequence evaluates all of them (nmgp_svc_batch_*), which is
what amortises the latency-bound panel steps of the Cholesky.  ``value`` counts evaluations: steps x chains x GPUs /
time.  ``--chains 1`` gives the single-chain latency path.  Data follow the reference simulator's recipe
(SIM_code/sim.py:177-263); the subject's data and all parameter vectors are resident in HBM when the timed region
starts, and after every step the host reads back what an MCMC driver needs: the verbose scalars and status of EVERY
chain (and, with ``--grad``, every chain's gradient).  With N GPUs every rank owns its own subject and chains: weak
scaling, no collective on the data path; RCCL is used for the barrier, the max-over-ranks time and the final reduction of
the chains' statistics only.

``--workload separable`` is BASELINE config 5 under the same contract: the separable model (logpos.py:216-296) at N = 4096, D = 5,
``--chains`` (default 16) chains per step through nmgp_sep_batch_eval (the chains' B x D blocks form one batch of the blocked
Cholesky); ``--workload subjects`` is config 4 (independent subjects, one multi-subject batch per GPU).

One JSON line is printed by rank 0 (contract in the task statement).  Flat scalars come first so that a truncated record still
shows them (value_grad_evals_per_s, hmc_samples_per_s, roofline_frac_end_to_end, ...), then these objects:
  roofline      -- the dominant kernel (k_syrk_lower, the FP64-MFMA trailing update of the blocked Cholesky): algorithmic
                   flop of its launches divided by their summed HIP-event durations on the launching stream.
  grad          -- the MCMC-relevant rate: value+gradient evaluations / second at the same chain count (HMC spends 20
                   leapfrog GRADIENT evaluations per sample, Nonseparable_model.py:228-231), with its own ms_per_step and
                   its end-to-end roofline on n^3 flop per evaluation (W_fb = n^3).
  hmc           -- (single-GPU lines) BatchedHMC at the same chain count: 5 samples of 20 leapfrog steps under the prior-factor metric
                   (drivers.PriorMetric; --hmc-mass identity,... for the others) from committed typical-set positions: samples/s,
                   gradient evaluations/s, acceptance, and where a sample's wall time goes (device_share).
  cpu_baseline  -- the NumPy/SciPy oracle (oracle/nmgp_oracle.py) timed on this host's cores on a bounded sample of the
                   same workload by rank 0 AFTER the timed region (the other ranks wait at the final barrier; fewer evaluations
                   when N > 1): Cholesky formulation, and the reference's own inverse+logdet formulation (logpos.py:352-353)
                   beside it.

The control flow (partition -> warm-up -> barrier -> K timed steps -> barrier -> max over ranks -> one reduction) is
written against a small backend interface so that tests/test_bench_flow_gloo.py can drive the SAME code with two gloo ranks
and the CPU oracle as evaluator; the product backend below is the only one bench.py itself can construct.
"""
import argparse
import hashlib
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
SVC_KEYS = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")
CHOL_SOURCE = os.path.join(ROOT, "nonstationary_multivariate_gaussian_process_amd", "csrc", "nmgp_chol.hip")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


SCHEDULE_NEUTRAL_ENV = ("NMGP_ROUND", "NMGP_BENCH_SELF_LAUNCHED", "NMGP_POISON")


def measured_traffic(N, M, chains, want_grad=False, workload="chain"):
    """HBM bytes of one step as measured by the rocprofv3 PMC passes whose summary is committed under profiles/ (FETCH_SIZE x2
    for gfx950 + WRITE_SIZE, MI355X_MICROARCH.md) -> (bytes, note, scope).  `scope` names the kernels the figure covers
    ("k_syrk_lower launches" for the value line, whose roofline names that kernel; "every kernel of the evaluation" for the
    end-to-end rooflines) and goes into the JSON as `traffic_scope`.  The figure belongs to the sources it was measured on: an
    entry whose scope is k_syrk_lower alone is pinned to the SHA-256 of csrc/nmgp_chol.hip, a whole-evaluation entry to the
    library's build id (build.tree_id(): every source, header and code-generation flag) -- otherwise null, with the reason.  Null
    as well when a schedule-changing NMGP_* environment variable is set: the committed passes ran the default schedule."""
    switches = sorted(k for k in os.environ if k.startswith("NMGP_") and k not in SCHEDULE_NEUTRAL_ENV)
    if switches:
        return None, "not reported: %s set (the committed PMC passes ran the default schedule)" % ", ".join(switches), None
    try:
        with open(TRAFFIC_FILE) as f:
            entries = json.load(f)["entries"]
        with open(CHOL_SOURCE, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        from nonstationary_multivariate_gaussian_process_amd import build
        tree = build.tree_id()
    except Exception as e:      # noqa: BLE001
        return None, "no committed PMC measurement (%s)" % type(e).__name__, None
    for e in entries:
        if (e["N"], e["M"], e["chains"], bool(e.get("grad", False)), e.get("workload", "chain")) == (
                N, M, chains, bool(want_grad), workload):
            whole = e.get("scope") != "k_syrk_lower launches"
            if whole and e.get("tree_id") != tree:
                return None, "stale: %s was measured on another build of the library (whole-evaluation figures are pinned to " \
                             "build.tree_id())" % e["source"], e.get("scope")
            if not whole and e["chol_sha256"] != sha:
                return None, "stale: %s was measured on another revision of csrc/nmgp_chol.hip" % e["source"], e.get("scope")
            return float(e["bytes_per_step"]), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s" % e["source"], e.get("scope")
    return None, "no committed PMC measurement for N=%d, M=%d, %d %s%s" % (
        N, M, chains, workload, ", value+gradient" if want_grad else ""), None


# ---- product backend: libnmgp_hip.so on one MI355X per rank, RCCL between ranks -------------------------------------
class HipBackend:
    """One process per GPU.  Everything that touches HIP / torch.cuda / RCCL lives here.

    `rehearse` (bench.py --rehearse-on-one-gpu): the multi-process path on a box with FEWER GPUs than ranks -- rank r drives GPU
    r mod visible, the process group is gloo and the few control-plane collectives run on CPU tensors (RCCL refuses two ranks on
    one device).  Everything else is the product path: N processes launched the same way, each loading libnmgp_hip.so, creating
    its context and streams and running the real evaluations.  The line says `rehearsal`, and its value is not a scaling
    figure (the ranks share a card)."""
    device = "cuda"

    def __init__(self, local_rank, rehearse=False):
        import torch
        self.torch = torch
        self.rehearse = bool(rehearse)
        ndev = torch.cuda.device_count()
        if ndev <= 0:
            raise SystemExit("bench.py: no GPU is visible (the MI355X path has no CPU fallback)")
        if local_rank >= ndev and not rehearse:
            raise SystemExit("bench.py: LOCAL_RANK=%d but this process sees %d GPU(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES "
                             "= %r / %r): one rank per GPU needs --nproc-per-node <= visible GPUs" % (
                                 local_rank, ndev, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
        self.rank_local = local_rank
        self.local_rank = local_rank % ndev          # the device ordinal this rank drives
        if rehearse:
            self.device = "cpu"                      # where the control-plane collectives' tensors live
        torch.cuda.set_device(self.local_rank)

    def device_count(self):
        return self.torch.cuda.device_count()

    def init_dist(self, rank, world):
        import torch.distributed as dist
        if self.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=self.torch.device("cuda", self.local_rank))

    def sync(self):
        self.torch.cuda.synchronize()

    def identity(self):
        """Which GPU this rank really drives: torch's current device, its PCI address and UUID, and the device the HIP
        library itself reports as current (both must be LOCAL_RANK) -- first-contact evidence for a multi-GPU run."""
        t = self.torch
        pr = t.cuda.get_device_properties(self.local_rank)
        pci = None
        if hasattr(pr, "pci_bus_id"):
            pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        from nonstationary_multivariate_gaussian_process_amd import _lib
        return {"local_rank": self.rank_local, "device_ordinal": self.local_rank, "torch_current_device": int(t.cuda.current_device()), "name": pr.name,
                "pci_bus_id": pci, "uuid": str(getattr(pr, "uuid", "")), "pid": os.getpid(),
                "visible_devices": int(t.cuda.device_count()), "library_build_id": _lib.build_id()}

    def chains(self, d, allp, hv, groups):
        return HipChains(self.local_rank, d, allp, hv, groups)

    def subjects(self, subs, pars, hv, chains_per_subject=1):
        return HipSubjects(self.local_rank, subs, pars, hv, chains_per_subject)

    def separable(self, d, pars, hv):
        return HipSeparable(self.local_rank, d, pars, hv)


class HipChains:
    """B chains of one subject: one batched context per stream group (or the resident single-chain entry for B = 1)."""

    def __init__(self, device, d, allp, hv, groups):
        from nonstationary_multivariate_gaussian_process_amd import _lib
        self.hv = hv
        self.B = allp.shape[0]
        G = max(1, min(groups, self.B))
        self.ctxs = [_lib.Context(device) for _ in range(G)]
        self.ctx = self.ctxs[0]
        self.sizes = [self.B // G + (1 if g < self.B % G else 0) for g in range(G)]
        off = 0
        for cg, sz in zip(self.ctxs, self.sizes):
            cg.set_data(d["x"], d["Y"])
            if self.B > 1:
                cg.svc_batch_alloc(sz)
                cg.svc_batch_set_pars(allp[off:off + sz])
            else:
                cg.svc_set_pars(allp[0])
            off += sz
        self.grads = None

    def step(self, want_grad):
        """One evaluation of every chain; returns (out [B, 5], status [B]) and keeps the gradients [B, P] in self.grads."""
        if self.B == 1:
            self.ctx.svc_eval_resident(self.hv, True, want_grad)
            out, g = self.ctx.svc_fetch(want_grad)
            self.grads = g[None, :] if want_grad else None
            return out[None, :], np.zeros(1, dtype=np.int32)
        for cg in self.ctxs:
            cg.svc_batch_eval(self.hv, True, want_grad)
        outs, sts, gs = [], [], []
        for cg in self.ctxs:
            o, st = cg.svc_batch_fetch()
            outs.append(o)
            sts.append(st)
            if want_grad:
                gs.append(cg.svc_batch_fetch_grad())
        self.grads = np.concatenate(gs) if want_grad else None
        return np.concatenate(outs), np.concatenate(sts)

    def sync(self):
        for cg in self.ctxs:
            cg.sync()

    def close(self):
        for cg in self.ctxs:
            cg.close()


class HipSubjects:
    """The rank's subjects as ONE multi-subject batch (own x, Y and prior factors per batch element)."""

    def __init__(self, device, subs, pars, hv, chains_per_subject=1):
        from nonstationary_multivariate_gaussian_process_amd import _lib
        self.hv = hv
        self.ctx = _lib.Context(device)
        self.ctx.set_data(subs[0]["x"], subs[0]["Y"])
        self.ctx.svc_batch_alloc(len(subs) * chains_per_subject)
        self.ctx.svc_batch_set_subjects(np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs]), chains_per_subject)
        self.ctx.svc_batch_set_pars(pars)
        self.grads = None

    def step(self, want_grad):
        self.ctx.svc_batch_eval(self.hv, True, want_grad)
        out, st = self.ctx.svc_batch_fetch()
        self.grads = self.ctx.svc_batch_fetch_grad() if want_grad else None
        return out, st

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.ctx.close()


class HipSeparable:
    """B chains of the SEPARABLE model of one subject: one nmgp_sep_batch_eval per step (the chains' B x D blocks are one batch of the
    blocked Cholesky).  The entry takes the parameter vectors from the host every call -- they are what an MCMC driver hands over
    per leapfrog step (B x (2N + T + 1) doubles: 1 MB at the config's size)."""

    def __init__(self, device, d, pars, hv):
        from nonstationary_multivariate_gaussian_process_amd import _lib
        self.hv = hv
        self.pars = np.ascontiguousarray(pars)
        self.ctx = _lib.Context(device)
        self.ctx.set_data(d["x"], d["Y"])
        self.grads = None

    def step(self, want_grad):
        out, self.grads, st = self.ctx.sep_batch_eval(self.pars, self.hv, True, want_grad)
        return out, np.where(st >= 0, 0, st)          # (1..3 = evaluated with jitter retries: a valid evaluation)

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.ctx.close()


# ---- backend-independent control flow -------------------------------------------------------------------------------
def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


DIST_ACTIVE = False      # a process group is initialised (world > 1, or one rank under --dist-selftest)


def _dist_on(world):
    return world > 1 or DIST_ACTIVE


def barrier(be, ev, world):
    """The bracket of the timed region: all ranks arrive, all device work of this rank is complete."""
    if _dist_on(world):
        import torch.distributed as dist
        dist.barrier()
    be.sync()
    ev.sync()


def max_over_ranks(elapsed, world, device):
    if not _dist_on(world):
        return elapsed
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def per_rank(value, world, device):
    """The same scalar of every rank, in rank order (first-contact diagnostics of a multi-GPU run: which rank is the slow one)."""
    if not _dist_on(world):
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o[0]) for o in out]


RANK_IDS = None          # every rank's identity() in rank order, gathered once after the process group is up


def gather_identities(be, rank, world):
    """All ranks: collect {rank, local_rank, device, PCI address, pid, ...} of every rank (one all-gather of small objects)."""
    ident = getattr(be, "identity", None)
    mine = dict(ident() if ident else {"local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()}, rank=rank)
    if not _dist_on(world):
        return [mine]
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    return out


def dist_report(be, world, rank_seconds, steps):
    """What the process group looked like from rank 0, every rank's device, and the spread of the ranks' own step times (the
    headline uses the max)."""
    rep = {"world_size_env": world, "self_launched": os.environ.get("NMGP_BENCH_SELF_LAUNCHED") == "1",
           "ms_per_step_by_rank": [1e3 * t / max(steps, 1) for t in rank_seconds],
           "ms_per_step_min_rank": 1e3 * min(rank_seconds) / max(steps, 1),
           "ms_per_step_max_rank": 1e3 * max(rank_seconds) / max(steps, 1)}
    dc = getattr(be, "device_count", None)
    rep["gpus_visible_to_rank0"] = dc() if dc else None
    if _dist_on(world):
        import torch.distributed as dist
        rep["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank()}
    if RANK_IDS is not None:
        rep["ranks"] = RANK_IDS
        pcis = [r.get("pci_bus_id") for r in RANK_IDS if r.get("pci_bus_id")]
        rep["distinct_gpus"] = len(set(pcis)) if pcis else None      # must equal the world size on a real multi-GPU run
    return rep


def timed_steps(be, ev, world, steps, want_grad):
    """EXACTLY `steps` steps bracketed by barrier + device synchronisation on both sides; the max over ranks."""
    out = status = None
    barrier(be, ev, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        out, status = ev.step(want_grad)
    barrier(be, ev, world)
    mine = time.perf_counter() - t0
    timed_steps.rank_seconds = per_rank(mine, world, be.device)      # kept for the `distributed` object of the JSON line
    return max_over_ranks(mine, world, be.device), out, status


def unit_rows(ids, steps, outs, status):
    """One row per unit (chain or subject): [global id, ok, evals, NegLog, loglik, lp_l, lp_uL, lp_s2]."""
    return np.array([[i, float(st == 0), steps] + [float(v) for v in o[:5]] for i, o, st in zip(ids, outs, status)])


def map_point(N, M, seed):
    """The committed MAP estimate of the subject (N, M, data seed), or None."""
    path = os.path.join(ROOT, "tests", "golden", "map_N%d_M%d_seed%d.npz" % (N, M, seed))
    if not os.path.exists(path):
        return None
    g = dict(np.load(path))
    g["path"] = path
    return g


def hmc_state(N, M, seed):
    """The committed sampler state of the subject (N, M, data seed): polished MAP point + typical-set positions, or None."""
    path = os.path.join(ROOT, "tests", "golden", "hmc_state_N%d_M%d_seed%d.npz" % (N, M, seed))
    if not os.path.exists(path):
        return None
    g = dict(np.load(path))
    g["path"] = path
    return g


def chain_parameters(sim, d, B):
    """Chain b starts from its own smooth perturbation of the generating parameters."""
    return np.stack([sim.perturb(d["pars_true"], 0.05, 0.7 + 0.37 * b) for b in range(B)])


def hmc_measure(a, kind, rank, world, be, ev, prof, d, hyper, allp, B):
    """End-to-end MCMC rate: BatchedHMC (drivers.py) advances B chains in lock-step, 20 leapfrog steps per sample
    (Nonseparable_model.py:228-231), every step one batched value+gradient launch sequence + the leapfrog kernels.  Like the
    reference's sampler the chains start from the MAP estimate when one is committed for this rank's subject
    (tests/golden/map_N2048_M3_seed2222.npz, made by tools/make_map_point.py) -- every chain from the same point with its own
    momenta -- so that acceptance and energy error are sampler evidence, not the descent of a perturbed start.
    kind: prior (the prior-factor metric, drivers.PriorMetric: cached GP-prior factors + a low-rank likelihood correction built here
    from batched Hessian-vector products; the metric under which the N = 2048 chains mix, profiles/r05_hmc_1000.json), identity
    (the call of Nonseparable_model.py:228-231 as written), diag / dense (a synthetic matrix of that shape: rate of the device path)."""
    from nonstationary_multivariate_gaussian_process_amd import drivers
    N, M = a.N, a.M
    q0, start = allp, "the chains' own perturbations of the generating parameters (no MAP point committed for this subject)"
    mp = map_point(N, M, 2222 + rank)
    if mp is not None:
        q0 = np.repeat(mp["pars_map"][None, :], B, axis=0)
        start = ("the MAP estimate of this subject (tests/golden/%s: %d Adam iterations at lr %.1f by tools/make_map_point.py, "
                 "log posterior %.3f), every chain with its own momenta" % (
                     os.path.basename(mp["path"]), int(mp["iterations"]), float(mp["lr"]), float(mp["target_value_hist"][-1])))
    mass_kw, mass_note, step = {}, "identity mass matrix", a.hmc_step
    metric_rec = None
    if kind == "prior":
        # the metric is built at the MODE (the Adam estimate polished by L-BFGS) and the chains start from typical-set positions of
        # a previous run under this metric -- both committed for rank 0's subject (tests/golden/hmc_state_*.npz, written by
        # tools/hmc_1000.py --save-state); chains started AT the mode would spend their first trajectories converting P/2 units
        # of kinetic energy, whose leapfrog error rejects any step worth having (profiles/r05_hmc_1000.json: warm_up)
        st = hmc_state(N, M, 2222 + rank)
        q_ref = q0[0]
        step = a.hmc_prior_step
        if st is not None:
            q_ref = st["pars_polished"]
            typ = st["pars_typical"]
            q0 = np.stack([typ[b % typ.shape[0]] for b in range(B)])
            start = ("typical-set positions of the %d chains of tools/hmc_1000.py's run under this metric (tests/golden/%s; chain b starts "
                     "from position b mod %d with its own momenta), metric built at the polished MAP point" % (
                         typ.shape[0], os.path.basename(st["path"]), typ.shape[0]))
        else:
            step = min(step, 0.03)          # from the (unpolished) MAP point itself: a thermalising step
        t0 = time.perf_counter()
        met = drivers.prior_lowrank_metric(d["x"], d["Y"], hyper, q_ref, rank=a.hmc_rank, oversample=32, power_iters=1, seed=7, ctx=prof,
                                           batch=B)
        metric_rec = dict({k: v for k, v in met.info.items() if k != "eigenvalues"}, seconds=time.perf_counter() - t0, rank=met.rank)
        mass_kw = {"M": met}
        mass_note = ("prior-factor metric M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T (cached GP-prior Cholesky factors + rank-%d "
                     "likelihood correction; whitened momenta, no [P, P] matrix)" % met.rank)
    elif kind != "identity":
        # a synthetic constant mass matrix of the right SHAPE (rate measurement of the diagonal / dense device path; the
        # reference derives M = inv(sample covariance) from a previous run, Nonseparable_model_mpiKAISER.py:398-411): M^-1 = s I
        # (+ a rank-8 term for "dense"), so that the trajectory in q is the identity-mass one at step sqrt(s) eps
        Pn = q0.shape[1]
        sc = 4.0
        if kind == "diag":
            mass_kw = {"Minv": np.full(Pn, sc)}
        else:
            rng = np.random.default_rng(5)
            W = rng.standard_normal((Pn, 8)) / np.sqrt(Pn)
            Minv = sc * np.eye(Pn) + sc * (W @ W.T)
            Mm = (np.eye(Pn) - W @ np.linalg.solve(np.eye(8) + W.T @ W, W.T)) / sc          # Woodbury
            mass_kw = {"M": Mm, "Minv": Minv}
        mass_note = "%s mass matrix (synthetic, M^-1 ~ %g I), resident on the device" % (kind, sc)
        step = a.hmc_step / 2.0
    hmc = drivers.BatchedHMC(d["x"], d["Y"], hyper, q0, step_size=step, num_steps_in_leap=20, seed=1, ctx=prof, device_momenta=True,
                             **mass_kw)
    barrier(be, ev, world)
    t0 = time.perf_counter()
    samples, info = hmc.run(a.hmc_samples)
    barrier(be, ev, world)
    h_elapsed = max_over_ranks(time.perf_counter() - t0, world, be.device)
    evals = (1 + 20 * a.hmc_samples) * B
    ee = info["energy_error"]
    moved = float(np.sqrt(np.mean((samples[-1] - q0) ** 2)))
    # every rank's own sampler evidence (ranks other than 0 have no committed MAP point: their chains start from perturbed
    # parameters -- descent trajectories, always accepted -- and the record must not pass rank 0's figures off as theirs)
    mine = {"rank": rank, "start": "MAP estimate" if mp is not None else "perturbed generating parameters (no MAP point committed)",
            "accept_rate_mean": float(np.mean(info["accept_rate"])), "accept_rate_min": float(np.min(info["accept_rate"])),
            "median_abs_energy_error": float(np.nanmedian(np.abs(ee))), "max_abs_energy_error": float(np.nanmax(np.abs(ee))),
            "rms_displacement_per_parameter": moved}
    by_rank = [mine]
    if _dist_on(world):
        import torch.distributed as dist
        by_rank = [None] * dist.get_world_size()
        dist.all_gather_object(by_rank, mine)
    tm = info.get("timing", {})
    rec = {"what": "BatchedHMC: %d chains in lock-step, %d samples per chain, 20 leapfrog steps per sample, %s (the sampler call "
                   "of Nonseparable_model.py:228-231); one batched value+gradient evaluation per leapfrog step; positions, momenta "
                   "and gradients stay in HBM for the whole trajectory (nmgp_svc_batch_traj / _traj_z), per sample the momenta's "
                   "standard normals go up and the end point comes down" % (B, a.hmc_samples, mass_note),
           "mass": kind, "step_size": step, "reference_step_size": 1e-4,
           "step_size_note": "the reference's 1e-4 (Nonseparable_model.py:229) is rejected every time from the MAP point at this "
                             "size under the identity (P = 14,337: energy error +8.0, profiles/r03_hmc_steps.txt); identity default "
                             "here 4e-5; under the prior-factor metric the posterior is ~N(0, I) and the step is O(P^-1/4)",
           "long_run": "profiles/r05_hmc_1000.json: BASELINE config 3 as worded -- 1000 iterations at this size (tools/hmc_1000.py)",
           "start": start,
           "samples_per_s": a.hmc_samples * B * world / h_elapsed, "samples_per_chain": a.hmc_samples,
           "seconds": h_elapsed, "grad_evals_per_s": evals * world / h_elapsed,
           # where a sample's wall time goes on rank 0: inside the synchronous trajectory calls (upload of the standard normals,
           # every leapfrog launch, the end point's download) against the host's own work around them (drawing B x P normals,
           # energies, accept test, bookkeeping)
           "device_share": tm.get("device_share"), "trajectory_call_seconds": tm.get("trajectory_call_seconds"),
           "host_seconds_in_the_loop": (tm.get("loop_seconds", 0.0) - tm.get("trajectory_call_seconds", 0.0)) if tm else None,
           "setup_seconds": tm.get("setup_seconds"),
           "accept_rate_mean": float(np.mean([r["accept_rate_mean"] for r in by_rank])),
           "accept_rate_min": float(np.min([r["accept_rate_min"] for r in by_rank])),
           "median_abs_energy_error": mine["median_abs_energy_error"], "max_abs_energy_error": mine["max_abs_energy_error"],
           "rms_displacement_per_parameter": moved, "by_rank": by_rank}
    if metric_rec is not None:
        rec["metric"] = metric_rec
    return rec


def run_chains(a, rank, world, be):
    """Headline workload: every rank owns one subject (seed 2222 + rank; 2222 is the reference's single-subject seed,
    sim.py:359) and B independent chains of it."""
    from nonstationary_multivariate_gaussian_process_amd import chains, sim
    N, M, B = a.N, a.M, max(1, a.chains)
    n = N * M
    d = sim.simulate_nonseparable(N, M, seed=2222 + rank)
    hyper = sim.HYPER_SVC
    hv = np.array([hyper[k] for k in SVC_KEYS], dtype=np.float64)
    allp = chain_parameters(sim, d, B)
    ev = be.chains(d, allp, hv, a.groups)
    want_grad = bool(a.grad)
    prof = getattr(ev, "ctx", None)          # stage timers exist on the HIP backend only
    if prof is not None:
        prof.profile_enable(True)   # on from the warm-up on: the HIP events are created on first use, outside the timed region
    for _ in range(a.warmup):
        ev.step(want_grad)
    if prof is not None:
        prof.profile_reset()
    elapsed, out, status = timed_steps(be, ev, world, a.steps, want_grad)
    rank_seconds = timed_steps.rank_seconds
    stage = prof.profile_read() if prof is not None else {}
    kprof = None
    if prof is not None:
        # dominant-kernel pass: the same K steps again with one HIP-event pair around EVERY k_syrk_lower launch (on the
        # stream it is launched on); kept out of the timed region so that the ~200 extra event records per factorisation
        # do not leak into `value`
        prof.profile_enable(2)
        prof.profile_reset()
        for _ in range(a.steps):
            ev.step(want_grad)
        barrier(be, ev, world)
        kprof = prof.profile_read_work()
        prof.profile_enable(False)
    # the MCMC-relevant rate: value + gradient at the same chain count (skipped when the headline itself is --grad)
    grad_rec = None
    if not want_grad and a.grad_steps > 0:
        for _ in range(max(1, min(a.warmup, 1))):
            ev.step(True)
        if prof is not None:
            prof.profile_enable(True)
            prof.profile_reset()
        g_elapsed, g_out, g_status = timed_steps(be, ev, world, a.grad_steps, True)
        g_stage = prof.profile_read() if prof is not None else {}
        if prof is not None:
            prof.profile_enable(False)
        g_rate = a.grad_steps * world * B / g_elapsed
        g_tf = g_rate * float(n) ** 3 / 1e12 / world
        gnorm = float(np.linalg.norm(ev.grads[0])) if ev.grads is not None else None
        g_traffic, g_traffic_note, g_scope = measured_traffic(N, M, B, True) if prof is not None else (None, "no HIP backend", None)
        grad_rec = {"what": "nlogpos_obj_SVC value + gradient of every chain per step (gradients copied to the host "
                            "every step), same %d chain(s) per GPU" % B,
                    "value": g_rate, "unit": "evals/s", "steps": a.grad_steps, "ms_per_step": 1e3 * g_elapsed / a.grad_steps,
                    "chains_ok": int(np.sum(g_status == 0)), "grad_norm_chain0": gnorm,
                    "roofline": {"what": "end to end: n^3 flop per value+gradient evaluation (SURVEY 8d W_fb) x evals/s "
                                         "per GPU", "bound": "mfma", "achieved": g_tf, "peak": FP64_MATRIX_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": g_tf / FP64_MATRIX_PEAK_TFLOPS,
                                 "traffic": g_traffic, "traffic_note": g_traffic_note, "traffic_scope": g_scope},
                    "stage_ms": {k: (v[0] / max(v[1], 1)) for k, v in g_stage.items() if v[1] > 0}}
    hmc_rec, hmc_other = None, []
    hmc_on = a.hmc_samples > 0 and prof is not None and max(1, min(a.groups, B)) == 1 and B > 1
    if hmc_on and world > 1 and not a.hmc_all_ranks:
        hmc_on = False          # a multi-GPU line measures the evaluations; the sampler object belongs to the N = 1 line
    if hmc_on:
        for kind in [k.strip() for k in a.hmc_mass.split(",") if k.strip()]:
            r = hmc_measure(a, kind, rank, world, be, ev, prof, d, hyper, allp, B)
            if hmc_rec is None:
                hmc_rec = r
            else:
                hmc_other.append(r)
        if hmc_other:
            hmc_rec["other_mass_matrices"] = hmc_other
    # the ONE reduction: every chain of every rank contributes a row
    ids = [rank * B + b for b in range(B)]
    stats, table = chains.reduce_rows(unit_rows(ids, a.steps, out, status), B * world, world, device=be.device)
    rec = None
    if rank == 0:
        value = a.steps * world * B / elapsed
        rec = {
            "metric": "log-posterior evals/sec (N=2048, D=3 nonseparable GP)" if (N, M) == (2048, 3) else
                      "log-posterior evals/sec (N=%d, D=%d nonseparable GP)" % (N, M),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "nonseparable GP nlogpos_obj_SVC %s, D=%d, N=%d (MN=%d), %d independent chain(s) per "
                                   "GPU evaluated per step" % ("value+gradient" if want_grad else "value", M, N, n, B),
                       "chains_per_gpu": B, "stream_groups": max(1, min(a.groups, B)),
                       "host_reads_per_step": "verbose scalars + status of every chain" +
                                              (", gradients [B, P] of every chain" if want_grad else ""),
                       "neglog_rank0_chain0": float(out[0][0]),
                       "chains_total": B * world, "chains_ok": int(stats[0]), "chains_failed": int(stats[1]),
                       "sum_neglog_all_chains": float(stats[3]), "chain_table_rows": int(table.shape[0])},
        }
        if prof is not None:
            rec.update(hip_chain_report(a, prof, stage, kprof, B, n, want_grad))
        if grad_rec is not None:
            rec["grad"] = grad_rec
        if hmc_rec is not None:
            rec["hmc"] = hmc_rec
        elif world > 1 and a.hmc_samples > 0:
            rec["hmc_note"] = "not measured on multi-GPU lines (--hmc-all-ranks turns it on): the N = 1 line carries the sampler object"
        rec["distributed"] = dist_report(be, world, rank_seconds, a.steps)
        if not a.no_cpu_baseline:
            # every line carries its CPU figure: rank 0 times the oracle AFTER the timed region (the other ranks wait at the final
            # barrier), with fewer evaluations when N > 1 so that a multi-GPU run is not held up
            ce, cg = (a.cpu_evals, a.cpu_grad_evals) if world == 1 else (min(a.cpu_evals, 3), min(a.cpu_grad_evals, 1))
            rec["cpu_baseline"] = cpu_baseline(d, allp[0], hyper, ce, want_grad, cg)
        # flat scalars first: a truncated record still shows them
        e2e = value * float(n) ** 3 / (1.0 if want_grad else 3.0) / 1e12 / world / FP64_MATRIX_PEAK_TFLOPS
        flat = {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": world,
                "roofline_frac_end_to_end": e2e,
                "roofline_frac_kernel": rec.get("roofline", {}).get("frac"),
                "value_grad_evals_per_s": value if want_grad else (grad_rec["value"] if grad_rec else None),
                "value_grad_roofline_frac_end_to_end": e2e if want_grad else (grad_rec["roofline"]["frac"] if grad_rec else None),
                "hmc_samples_per_s": hmc_rec["samples_per_s"] if hmc_rec else None,
                "hmc_grad_evals_per_s": hmc_rec["grad_evals_per_s"] if hmc_rec else None,
                "hmc_device_share": hmc_rec["device_share"] if hmc_rec else None,
                "cpu_evals_per_s": rec["cpu_baseline"]["value"] if "cpu_baseline" in rec else None}
        flat.update({k: v for k, v in rec.items() if k not in flat})
        rec = flat
    ev.close()
    return rec, stats, table


def hip_chain_report(a, ctx, stage, kprof, B, n, want_grad):
    """roofline object and measured-peak context from the HIP-event timers of rank 0's first stream group."""
    N, M = a.N, a.M
    G = max(1, min(a.groups, B))
    chol_ms, chol_cnt = stage["chol"]
    chol_avg_s = (chol_ms / max(chol_cnt, 1)) * 1e-3
    B0 = (B // G + (1 if B % G else 0)) if B > 1 else 1     # chains in the profiled context (group 0)
    flops = B0 * (2.0 if want_grad else 1.0) * float(n) ** 3 / 3.0
    chol_tf = flops / chol_avg_s / 1e12 if chol_avg_s > 0 else 0.0
    # k_syrk_lower: sum of algorithmic flop (2K per updated lower-trapezoid element) / sum of launch durations
    syrk_ms, syrk_cnt, syrk_flop, syrk_bytes = kprof["syrk"]
    achieved = syrk_flop / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
    syrk_info = {"launches_per_step": syrk_cnt / max(a.steps, 1), "avg_launch_us": 1e3 * syrk_ms / max(syrk_cnt, 1),
                 "ms_per_step": syrk_ms / max(a.steps, 1), "gflop_per_step": syrk_flop / max(a.steps, 1) / 1e9,
                 "algorithmic_bytes_per_step": syrk_bytes / max(a.steps, 1),
                 "share_of_factorisation_flop": syrk_flop / max(a.steps, 1) / flops}
    stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stage.items() if v[1] > 0}
    cov_ms = stage_ms.get("cov", 0.0)
    cov_bytes = B0 * 8.0 * n * (n + 1) / 2.0
    try:
        dgemm_tf = ctx.measure_dgemm_tflops(4096, 5)
        hbm = ctx.measure_hbm_rates(1 << 30, 10)
    except Exception:       # noqa: BLE001 -- measurement helpers are informative only
        dgemm_tf, hbm = None, {"copy": None, "read": None, "write": None}
    traffic, traffic_note, traffic_scope = measured_traffic(N, M, B, want_grad)
    return {
        "config_extra": {"stage_ms": stage_ms, "measured_dgemm_tflops_n4096": dgemm_tf, "measured_hbm_copy_gbs": hbm["copy"],
                         "measured_hbm_read_gbs": hbm["read"], "measured_hbm_write_gbs": hbm["write"],
                         "measured_hbm_note": "flat 16-byte-per-lane streaming kernels over 1 GiB buffers (tools/lab/hbm_lab.hip has "
                                              "the sweep; in-place read-modify-write of column-major panels tops out at ~5.1 TB/s, "
                                              "8-byte-per-lane panel writes at ~5.5 TB/s on the same chip: profiles/r03_hbm_lab*.txt)",
                         "cov_build_gbs": (cov_bytes / (cov_ms * 1e-3) / 1e9) if cov_ms > 0 else None},
        "roofline": {"kernel": "k_syrk_lower (v_mfma_f64_16x16x4_f64 trailing update of the blocked FP64 Cholesky of "
                               "%d %dx%d covariances): achieved = sum over launches of 2K*(updated lower-trapezoid "
                               "elements) / sum of HIP-event launch durations on the launching stream" % (B0, n, n),
                     "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                     "traffic_scope": traffic_scope,
                     "syrk": syrk_info,
                     "factorisation": {"what": "whole CHOL stage (syrk + potf2 + trsm + row moves), chains*n^3/3 flop%s "
                                               "over the HIP-event stage time of the timed region" % (
                                                   " x 2 (L^-T rows ride along)" if want_grad else ""),
                                       "ms": 1e3 * chol_avg_s, "achieved": chol_tf,
                                       "frac": chol_tf / FP64_MATRIX_PEAK_TFLOPS}},
    }


def run_subjects(a, rank, world, be):
    """BASELINE config 4 (Nonseparable_model_mpisim-style): independent subjects, D=M, N=--N (default there: 1024), subject s on
    rank s mod world (the reference maps rank -> data file, Nonseparable_model_mpisim.py:306).  The rank's subjects form ONE
    multi-subject batch: a step = one evaluation of EVERY local subject by one launch sequence.
    --subjects-total S (config 4 as worded: 64) fixes the job's subject count -- STRONG scaling, the per-GPU batch shrinks as GPUs
    are added; without it every GPU brings --subjects-per-gpu subjects of its own (weak scaling)."""
    from nonstationary_multivariate_gaussian_process_amd import chains, sim
    N, M = a.N, a.M
    strong = a.subjects_total > 0
    n_subj = a.subjects_total if strong else a.subjects_per_gpu * world
    if n_subj < world:
        raise SystemExit("bench.py: %d subject(s) cannot be sharded over %d ranks" % (n_subj, world))
    mine = chains.partition(n_subj, world, rank)
    hyper = sim.HYPER_SVC_MPISIM
    hv = np.array([hyper[k] for k in SVC_KEYS], dtype=np.float64)
    subs = [sim.simulate_nonseparable(N, M, seed=s_id) for s_id in mine]       # subject s uses seed s (sim.py:361-363)
    K = max(1, a.chains_per_subject)            # chains per subject in the rank's batch (subject-major)
    pars = np.stack([sim.perturb(d["pars_true"], 0.05, 0.7 + 0.37 * k) for d in subs for k in range(K)])
    ev = be.subjects(subs, pars, hv, K) if K > 1 else be.subjects(subs, pars, hv)
    want_grad = bool(a.grad)
    prof = getattr(ev, "ctx", None)
    if prof is not None:
        prof.profile_enable(True)
    for _ in range(a.warmup):
        ev.step(want_grad)
    if prof is not None:
        prof.profile_reset()
    elapsed, outs, status = timed_steps(be, ev, world, a.steps, want_grad)
    rank_seconds = timed_steps.rank_seconds
    stage = prof.profile_read() if prof is not None else {}
    if prof is not None:
        prof.profile_enable(False)
    unit_ids = [s_id * K + k for s_id in mine for k in range(K)]
    stats, table = chains.reduce_rows(unit_rows(unit_ids, a.steps, outs, status), n_subj * K, world, device=be.device)
    ev.close()
    rec = None
    if rank == 0:
        n = N * M
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stage.items() if v[1] > 0}
        chol_s = stage_ms.get("chol", 0.0) * 1e-3
        fact_tf = (len(mine) * K * (2.0 if want_grad else 1.0) * float(n) ** 3 / 3.0) / chol_s / 1e12 if chol_s > 0 else 0.0
        total = a.steps * n_subj * K
        value = total / elapsed
        traffic, traffic_note, traffic_scope = measured_traffic(N, M, len(mine) * K, want_grad, "subjects")
        e2e = value * float(n) ** 3 / (1.0 if want_grad else 3.0) / 1e12 / world / FP64_MATRIX_PEAK_TFLOPS
        rec = {
            "metric": "log-posterior evals/sec (%d subjects, N=%d, D=%d nonseparable GP)" % (n_subj, N, M),
            "value": value, "unit": "evals/s", "n_gpus": world, "roofline_frac_end_to_end": e2e,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "scaling_note": ("the job's %d subjects are fixed and sharded over the GPUs (config 4 as worded): the per-GPU batch shrinks "
                             "with N, and a batch of few subjects is latency-bound, so N GPUs give less than N x" % n_subj) if strong else
                            ("every GPU brings %d subjects of its own; config 4 as worded (a FIXED set of 64 subjects) is --subjects-total 64: "
                             "strong scaling" % a.subjects_per_gpu),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d independent subjects (%d on rank 0, one multi-subject batch per GPU%s), "
                                   "nlogpos_obj_SVC %s, D=%d, N=%d" % (n_subj, len(mine),
                                                                       ", %d chains per subject" % K if K > 1 else "",
                                                                       "value+gradient" if want_grad else "value", M, N),
                       "chains_per_subject": K,
                       "subjects_total": n_subj, "subjects_ok": int(stats[0]), "subjects_failed": int(stats[1]),
                       "sum_neglog": float(stats[3]), "subject_table_rows": int(table.shape[0]), "stage_ms": stage_ms},
            "roofline": {"kernel": "blocked FP64 Cholesky stage of rank 0's multi-subject batch (k_syrk_lower + panel "
                                   "kernels%s): subjects * n^3/3 flop%s over the HIP-event stage time" % (
                                       ", with the L^-T rows" if want_grad else "", " x 2" if want_grad else ""),
                         "bound": "mfma", "achieved": fact_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fact_tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_scope": traffic_scope},
            "distributed": dist_report(be, world, rank_seconds, a.steps)}
        if world == 1 and a.all_subjects > len(mine) and prof is not None:
            # the other end of config 4's scaling curve on the same box: ALL subjects of the 8-GPU job as one batch on this GPU
            rec["all_subjects_on_one_gpu"] = subjects_side_rate(a, be, a.all_subjects, hv, K, want_grad)
            rec["all_subjects_on_one_gpu"]["note"] = (
                "config 4 scales STRONGLY: %d subjects on one GPU run at this rate; %d GPUs x the per-GPU share's rate above is the "
                "most the sharded job can reach" % (a.all_subjects, max(1, a.all_subjects // max(len(mine), 1))))
        if not a.no_cpu_baseline:
            ce = a.cpu_evals if world == 1 else min(a.cpu_evals, 3)
            rec["cpu_baseline"] = cpu_baseline(subs[0], pars[0], hyper, ce, want_grad, 0)
            rec["cpu_evals_per_s"] = rec["cpu_baseline"]["value"]
    return rec, stats, table


def subjects_side_rate(a, be, S, hv, K, want_grad):
    """`S` subjects of the same recipe as ONE batch on this GPU: a few timed steps (rank 0 of a single-GPU run only)."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    subs = [sim.simulate_nonseparable(a.N, a.M, seed=s_id) for s_id in range(S)]
    pars = np.stack([sim.perturb(d["pars_true"], 0.05, 0.7 + 0.37 * k) for d in subs for k in range(K)])
    ev = be.subjects(subs, pars, hv, K) if K > 1 else be.subjects(subs, pars, hv)
    for _ in range(2):
        ev.step(want_grad)
    ev.sync()
    steps = max(3, a.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        out, st = ev.step(want_grad)
    ev.sync()
    dt = time.perf_counter() - t0
    ev.close()
    n = a.N * a.M
    rate = steps * S * K / dt
    return {"subjects": S, "value": rate, "unit": "evals/s", "ms_per_step": 1e3 * dt / steps, "subjects_ok": int(np.sum(st == 0)),
            "roofline_frac_end_to_end": rate * float(n) ** 3 / (1.0 if want_grad else 3.0) / 1e12 / FP64_MATRIX_PEAK_TFLOPS}


SEP_KEYS = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma", "beta_tilde_sigma", "a", "b", "c")


def run_separable(a, rank, world, be):
    """BASELINE config 5: the separable model (logpos.py:216-296; caller: Separable_model.py:160-166 MAP loop, :209-210 sampler) at
    N = 4096, D = 5.  Every rank owns one subject (seed 8 + rank) and B chains of it; a step = nlogpos_obj of every chain by one
    nmgp_sep_batch_eval (the chains' B x D blocks wB[p] K_x + sigma2 I as one batch of the blocked Cholesky).  Algorithmic work per
    evaluation: D N^3 / 3 flop (value), D N^3 (value+gradient: the L^-T rows and the inverse SYRK)."""
    from nonstationary_multivariate_gaussian_process_amd import chains, sim
    N, M, B = a.N, a.M, max(1, a.chains)
    d = sim.simulate_separable(N, M, 8 + rank)
    hyper = sim.HYPER_SEP
    hv = np.array([hyper[k] for k in SEP_KEYS], dtype=np.float64)
    pars = np.stack([sim.perturb(d["pars_true"], 0.05, 0.4 + 0.1 * k) for k in range(B)])
    ev = be.separable(d, pars, hv)
    want_grad = bool(a.grad)
    prof = getattr(ev, "ctx", None)
    if prof is not None:
        prof.profile_enable(True)
    for _ in range(a.warmup):
        ev.step(want_grad)
    if prof is not None:
        prof.profile_reset()
    elapsed, out, status = timed_steps(be, ev, world, a.steps, want_grad)
    rank_seconds = timed_steps.rank_seconds
    stage = prof.profile_read() if prof is not None else {}
    kprof = None
    if prof is not None:
        prof.profile_enable(2)          # dominant-kernel pass (see run_chains): one HIP-event pair per k_syrk_lower launch
        prof.profile_reset()
        for _ in range(a.steps):
            ev.step(want_grad)
        barrier(be, ev, world)
        kprof = prof.profile_read_work()
        prof.profile_enable(False)
    grad_rec = None
    if not want_grad and a.grad_steps > 0:
        ev.step(True)
        if prof is not None:
            prof.profile_enable(True)
            prof.profile_reset()
        g_elapsed, g_out, g_status = timed_steps(be, ev, world, a.grad_steps, True)
        g_stage = prof.profile_read() if prof is not None else {}
        if prof is not None:
            prof.profile_enable(False)
        g_rate = a.grad_steps * world * B / g_elapsed
        g_tf = g_rate * M * float(N) ** 3 / 1e12 / world
        grad_rec = {"what": "nlogpos_obj value + gradient of every chain per step (gradients returned to the host), same %d chain(s) "
                            "per GPU" % B,
                    "value": g_rate, "unit": "evals/s", "steps": a.grad_steps, "ms_per_step": 1e3 * g_elapsed / a.grad_steps,
                    "chains_ok": int(np.sum(g_status == 0)),
                    "grad_norm_chain0": float(np.linalg.norm(ev.grads[0])) if ev.grads is not None else None,
                    "roofline": {"what": "end to end: D N^3 flop per value+gradient evaluation x evals/s per GPU", "bound": "mfma",
                                 "achieved": g_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g_tf / FP64_MATRIX_PEAK_TFLOPS,
                                 "traffic": (measured_traffic(N, M, B, True, "separable") if prof is not None else (None, "no HIP backend", None))[0],
                                 "traffic_note": (measured_traffic(N, M, B, True, "separable") if prof is not None else (None, "no HIP backend", None))[1]},
                    "stage_ms": {k: (v[0] / max(v[1], 1)) for k, v in g_stage.items() if v[1] > 0}}
    ids = [rank * B + b for b in range(B)]
    stats, table = chains.reduce_rows(unit_rows(ids, a.steps, out, status), B * world, world, device=be.device)
    rec = None
    if rank == 0:
        value = a.steps * world * B / elapsed
        per_eval = M * float(N) ** 3 / (1.0 if want_grad else 3.0)
        e2e = value * per_eval / 1e12 / world / FP64_MATRIX_PEAK_TFLOPS
        stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in stage.items() if v[1] > 0}
        traffic, traffic_note, traffic_scope = measured_traffic(N, M, B, want_grad, "separable") if prof is not None else (
            None, "no HIP backend", None)
        rl = {"kernel": "k_syrk_lower", "bound": "mfma", "achieved": 0.0, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 0.0,
              "traffic": traffic, "traffic_note": traffic_note, "traffic_scope": traffic_scope,
              "algorithmic_bytes_per_step": B * M * 8.0 * N * (N + 1) / 2.0 * 2.0}        # the blocks written once, the factors read back once
        if kprof is not None:
            syrk_ms, syrk_cnt, syrk_flop, syrk_bytes = kprof["syrk"]
            ach = syrk_flop / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
            chol_s = stage_ms.get("chol", 0.0) * 1e-3
            fact_tf = B * M * (2.0 if want_grad else 1.0) * float(N) ** 3 / 3.0 / chol_s / 1e12 if chol_s > 0 else 0.0
            rl.update({"kernel": "k_syrk_lower (v_mfma_f64_16x16x4_f64 trailing update of the blocked FP64 Cholesky of the %d x %d = %d "
                                 "blocks wB[p] K_x + sigma2 I of order %d): achieved = sum over launches of 2K*(updated lower-trapezoid "
                                 "elements) / sum of HIP-event launch durations" % (B, M, B * M, N),
                       "achieved": ach, "frac": ach / FP64_MATRIX_PEAK_TFLOPS,
                       "syrk": {"launches_per_step": syrk_cnt / max(a.steps, 1), "ms_per_step": syrk_ms / max(a.steps, 1),
                                "gflop_per_step": syrk_flop / max(a.steps, 1) / 1e9},
                       "factorisation": {"what": "whole CHOL stage, chains x D x N^3/3 flop%s over the HIP-event stage time" % (
                                             " x 2 (L^-T rows ride along)" if want_grad else ""),
                                         "ms": 1e3 * chol_s, "achieved": fact_tf, "frac": fact_tf / FP64_MATRIX_PEAK_TFLOPS}})
        rec = {
            "metric": "log-posterior evals/sec (N=%d, D=%d separable GP)" % (N, M),
            "value": value, "unit": "evals/s", "n_gpus": world,
            "roofline_frac_end_to_end": e2e, "roofline_frac_kernel": rl["frac"],
            "value_grad_evals_per_s": value if want_grad else (grad_rec["value"] if grad_rec else None),
            "value_grad_roofline_frac_end_to_end": e2e if want_grad else (grad_rec["roofline"]["frac"] if grad_rec else None),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "separable GP nlogpos_obj %s, D=%d, N=%d (BASELINE config 5), %d independent chain(s) per GPU "
                                   "evaluated per step by one nmgp_sep_batch_eval (%d blocks of order %d per factorisation)" % (
                                       "value+gradient" if want_grad else "value", M, N, B, B * M, N),
                       "chains_per_gpu": B, "host_reads_per_step": "verbose tuples + status of every chain" +
                                                                   (", gradients [B, P] of every chain" if want_grad else ""),
                       "host_writes_per_step": "parameter vectors [B, 2N+T+1]",
                       "neglog_rank0_chain0": float(out[0][0]), "chains_total": B * world, "chains_ok": int(stats[0]),
                       "chains_failed": int(stats[1]), "sum_neglog_all_chains": float(stats[3]), "stage_ms": stage_ms},
            "roofline": rl}
        if grad_rec is not None:
            rec["grad"] = grad_rec
        rec["distributed"] = dist_report(be, world, rank_seconds, a.steps)
        if not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_separable(d, pars[0], hyper, 2 if world == 1 else 1, want_grad or grad_rec is not None)
            rec["cpu_evals_per_s"] = rec["cpu_baseline"]["value"]
    ev.close()
    return rec, stats, table


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(argv, n, script=None, out=None, timeout=None):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: THIS process -- which has imported
    neither torch nor the HIP library and has made no HIP call, and never will -- starts the N ranks as a CHILD
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same
    arguments>` (one rank per GPU over RCCL; a process that has touched the GPU is never re-exec'ed, a child is spawned
    and its exit code returned), relays rank 0's single JSON line to stdout and everything else the ranks print to
    stderr, and returns non-zero if any rank failed or the line count is not exactly one.  The reference's launch
    line for the same pattern is `mpirun -n 40 python Nonseparable_model_mpisim.py` (sim_job:9).
    `script` lets tests/test_bench_selflaunch.py point the ranks at a wrapper that calls bench.main with the gloo
    backend; bench.py itself always launches itself."""
    import subprocess
    out = out if out is not None else sys.stdout
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    env["NMGP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    print("bench.py: --gpus %d without WORLD_SIZE: launching the ranks as a child process: %s" % (n, " ".join(cmd)),
          file=sys.stderr, flush=True)
    # its own session = its own process group: on a timeout the whole group (launcher + ranks) is signalled, nothing else
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT, start_new_session=True)
    lines = []

    def stop_group():
        import signal
        for sig, grace in ((signal.SIGTERM, 20), (signal.SIGKILL, 20)):
            try:
                os.killpg(p.pid, sig)          # the group THIS call created (start_new_session): launcher and its workers
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=grace)
                # the launcher is gone; its workers get the same signal through the group -- give them a moment, then make sure
                time.sleep(0.5)
            except subprocess.TimeoutExpired:
                continue
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()

    def relay():
        for ln in p.stdout:
            is_rec = False
            if ln.startswith("{"):
                try:
                    is_rec = "metric" in json.loads(ln)
                except ValueError:
                    is_rec = False
            if is_rec:
                lines.append(ln)
            else:
                sys.stderr.write(ln)

    # The child's stdout is read on a thread so that `timeout` bounds the WHOLE run: a rank that hangs (RCCL initialisation, a
    # collective nobody joins) keeps the pipe open for ever, and a read loop on this thread would never reach the wait.
    import threading
    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    try:
        rc = p.wait(timeout=timeout)
        reader.join(timeout=30)
    except subprocess.TimeoutExpired:
        print("bench.py: the ranks did not finish within %s s: stopping the launcher child's process group (pgid %d)" % (timeout, p.pid),
              file=sys.stderr, flush=True)
        stop_group()
        reader.join(timeout=30)
        return 124
    except BaseException:
        stop_group()
        raise
    if rc != 0:
        print("bench.py: the torch.distributed.run child exited with %d (a rank failed)" % rc, file=sys.stderr, flush=True)
        return rc
    if len(lines) != 1:
        print("bench.py: expected exactly one JSON line from rank 0, got %d" % len(lines), file=sys.stderr, flush=True)
        return 1
    out.write(lines[0])
    out.flush()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=None, help="locations (default 2048; workload separable: 4096)")
    ap.add_argument("--M", type=int, default=None, help="outputs D (default 3; workload separable: 5)")
    ap.add_argument("--grad", action="store_true", help="time value+gradient evaluations as the headline")
    ap.add_argument("--grad-steps", type=int, default=3,
                    help="steps of the extra value+gradient measurement reported in the `grad` object (0 = skip)")
    ap.add_argument("--hmc-step", type=float, default=4e-5,
                    help="leapfrog step size of the `hmc` measurement.  The reference's call uses 1e-4 (Nonseparable_model.py:229); at "
                         "N = 2048 (P = 14,337) that step is rejected almost always from the MAP point (energy error +8.0), 4e-5 is "
                         "accepted 81 %% of the time (energy error 0.39): profiles/r03_hmc_steps.txt")
    ap.add_argument("--hmc-mass", default="prior",
                    help="mass matrix/matrices of the `hmc` measurement, comma-separated (the first is the `hmc` object, the others go "
                         "under hmc.other_mass_matrices): prior (default: the prior-factor metric, drivers.PriorMetric -- the one the "
                         "N = 2048 chains mix under), identity (the call of Nonseparable_model.py:228-231 as written), or a synthetic "
                         "diag / dense one resident on the device (nmgp_svc_batch_traj_set_mass; dense: P x P = 1.6 GB at the headline "
                         "size, one GEMM per leapfrog step for all chains; the host factors M once: ~1 min)")
    ap.add_argument("--hmc-rank", type=int, default=96, help="rank of the prior-factor metric's likelihood correction")
    ap.add_argument("--hmc-prior-step", type=float, default=0.08,
                    help="leapfrog step under the prior-factor metric (whitened coordinates: O(P^-1/4); 20 steps of 0.08 = a quarter period)")
    ap.add_argument("--hmc-all-ranks", action="store_true", help="measure the `hmc` object on multi-GPU runs too (every rank its own chains)")
    ap.add_argument("--hmc-samples", type=int, default=5,
                    help="samples per chain of the BatchedHMC end-to-end measurement reported in the `hmc` object (0 = skip)")
    ap.add_argument("--workload", choices=["chain", "subjects", "separable"], default="chain",
                    help="chain: B chains of one N=2048 subject per GPU (headline); subjects: BASELINE config 4, "
                         "independent subjects of size --N sharded round-robin over the GPUs (8 per GPU), one batch each; "
                         "separable: BASELINE config 5, B chains of the separable model at N=4096, D=5 per GPU")
    ap.add_argument("--subjects-per-gpu", type=int, default=8)
    ap.add_argument("--subjects-total", type=int, default=0,
                    help="workload subjects: a FIXED number of subjects for the whole job (config 4 as worded: 64), sharded over the GPUs "
                         "-- strong scaling; 0 (default): --subjects-per-gpu subjects per GPU -- weak scaling")
    ap.add_argument("--all-subjects", type=int, default=64,
                    help="workload subjects, single-GPU runs: also time this many subjects as ONE batch on the GPU (the other end of "
                         "config 4's scaling curve; 0 = skip)")
    ap.add_argument("--chains-per-subject", type=int, default=1,
                    help="workload subjects: chains per subject in the rank's batch (they share the subject's data and prior "
                         "factors on the device); a step then evaluates subjects x chains parameter vectors")
    ap.add_argument("--chains", type=int, default=None,
                    help="independent MCMC chains of the subject evaluated per step through the batched entry "
                         "(nmgp_svc_batch_* / nmgp_sep_batch_eval): one launch sequence covers all chains (default 128; separable: 16)")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the chains into this many groups, each a batched context on its own pair of HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run --gpus N ranks on a box with fewer GPUs: rank r drives GPU r mod visible, gloo process group, collectives "
                         "on CPU tensors (RCCL refuses two ranks on one device); exercises launch, per-rank contexts and the reduction "
                         "on hardware -- not a scaling measurement (the line carries `rehearsal`: true)")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="with ONE rank: initialise the process group anyway (RCCL) and run the barrier / max / gather / reduction "
                         "through it -- the part of the multi-GPU path a single-GPU box can execute")
    ap.add_argument("--cpu-evals", type=int, default=5, help="timed value evaluations of the CPU oracle (cpu_baseline; median)")
    ap.add_argument("--cpu-grad-evals", type=int, default=2,
                    help="timed value+gradient evaluations of the CPU oracle (cpu_baseline.grad, next to the `grad` object)")
    a = ap.parse_args(argv)
    sep = a.workload == "separable"
    if a.N is None:
        a.N = 4096 if sep else 2048
    if a.M is None:
        a.M = 5 if sep else 3
    if a.chains is None:
        a.chains = 16 if sep else 128
    for k in a.hmc_mass.split(","):
        if k.strip() not in ("prior", "identity", "diag", "dense"):
            ap.error("--hmc-mass: unknown kind %r" % k)
    return a


def main(argv=None, backend=None):
    """`backend` is a test hook (tests/test_bench_flow_gloo.py passes a gloo + CPU-oracle backend); bench.py run as a
    program always builds the HIP backend and fails loudly without a GPU."""
    a = parse_args(argv)
    rank, world, local_rank = dist_env()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and backend is None:
        # before anything touches HIP / torch.cuda: become the launcher of the N ranks (see self_launch)
        raise SystemExit(self_launch(sys.argv[1:] if argv is None else argv, a.gpus))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (`python bench.py --gpus N` does it by itself, "
                         "or `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)" % (a.gpus, world))
    be = backend if backend is not None else HipBackend(local_rank, rehearse=a.rehearse_on_one_gpu)
    global DIST_ACTIVE
    if world > 1 or a.dist_selftest:
        # --dist-selftest: ONE rank initialises the process group as well (RCCL on the HIP backend), so that the barrier, the
        # max-over-ranks, the per-rank gather and the final reduction run through torch.distributed in the same process as the
        # library's streams -- what a 1-GPU box can exercise of the multi-GPU path
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        be.init_dist(rank, world)
        DIST_ACTIVE = True
    global RANK_IDS
    RANK_IDS = gather_identities(be, rank, world)
    rec, stats, table = {"subjects": run_subjects, "separable": run_separable, "chain": run_chains}[a.workload](a, rank, world, be)
    if rank == 0:
        extra = rec.pop("config_extra", None)
        if extra:
            rec["config"].update(extra)
        if a.rehearse_on_one_gpu:
            rec["rehearsal"] = True
            rec["rehearsal_note"] = ("ranks share GPU(s) (rank r -> GPU r mod visible) and talk gloo: a functional rehearsal of the "
                                     "multi-process path, NOT a scaling measurement")
        if RANK_IDS and RANK_IDS[0].get("library_build_id"):
            rec["config"]["library_build_id"] = RANK_IDS[0]["library_build_id"]      # == build.tree_id() of the sources (_lib.load checks)
        print(json.dumps(rec), flush=True)
    if DIST_ACTIVE:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        DIST_ACTIVE = False
    return rec, stats, table


def cpu_baseline(d, pars, hyper, evals, want_grad, grad_evals=0):
    """The CPU oracle on the same subject: 1 warm-up + `evals` timed evaluations in the Cholesky formulation, then ONE
    timed evaluation in the reference's own formulation (dense inverse + logdet, logpos.py:352-353) so that the speed-up
    is not inflated by the reference's wasteful formulation.  `cores` = the BLAS/LAPACK threads NumPy/SciPy
    actually use on this host (threadpoolctl), else the affinity mask."""
    from oracle import nmgp_oracle as O     # checker / baseline only
    cores = len(os.sched_getaffinity(0))
    try:        # the threads the BLAS/LAPACK behind NumPy/SciPy actually runs (the elementwise parts are single-threaded)
        from threadpoolctl import threadpool_info
        blas = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = int(max(blas))
    except Exception:       # noqa: BLE001
        pass

    def one(form):
        return O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **hyper, verbose=True, formulation=form, grad=want_grad)
    one("cholesky")
    ts = []
    for _ in range(evals):
        t0 = time.perf_counter()
        one("cholesky")
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    t0 = time.perf_counter()
    one("reference")
    t_ref = time.perf_counter() - t0
    N, M = d["Y"].shape
    grad_rec = None
    if grad_evals > 0 and not want_grad:
        tg = []
        for _ in range(grad_evals):
            t0 = time.perf_counter()
            O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **hyper, verbose=True, formulation="cholesky", grad=True)
            tg.append(time.perf_counter() - t0)
        mg = float(np.median(tg))
        grad_rec = {"value": 1.0 / mg, "unit": "evals/s",
                    "sample": "%d value+gradient evaluations of the same subject (median %.2f s each), NumPy/SciPy oracle with "
                              "analytic adjoints, Cholesky formulation: the CPU figure beside the `grad` object" % (grad_evals, mg)}
    rec = _cpu_rec(med, cores, evals, N, M, want_grad, t_ref)
    if grad_rec is not None:
        rec["grad"] = grad_rec
    return rec


def _cpu_rec(med, cores, evals, N, M, want_grad, t_ref):
    return {"value": 1.0 / med, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d evaluations of the same N=%d, D=%d subject (median %.2f s each), NumPy/SciPy oracle, "
                      "Cholesky formulation, %s" % (evals, N, M, med, "value+gradient" if want_grad else "value only"),
            "reference_formulation": {"value": 1.0 / t_ref, "unit": "evals/s",
                                      "sample": "1 evaluation (%.2f s) of the same subject with the reference's dense "
                                                "inverse + logdet (logpos.py:352-353), NumPy/SciPy oracle" % t_ref}}


def cpu_baseline_separable(d, pars, hyper, evals, with_grad):
    """The CPU oracle's separable objective (nlogpos_obj: the reference's eigendecomposition formulation, distributions.py:26-52) on
    the same subject: 1 warm-up + `evals` timed evaluations (one N x N symmetric eigendecomposition each), and ONE timed
    value+gradient evaluation beside the `grad` object."""
    from oracle import nmgp_oracle as O     # checker / baseline only
    cores = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_info
        blas = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = int(max(blas))
    except Exception:       # noqa: BLE001
        pass
    N, M = d["Y"].shape
    O.nlogpos_obj(pars, d["Y"], d["x"], **hyper, verbose=True)
    ts = []
    for _ in range(max(1, evals)):
        t0 = time.perf_counter()
        O.nlogpos_obj(pars, d["Y"], d["x"], **hyper, verbose=True)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    rec = {"value": 1.0 / med, "unit": "evals/s", "cores": cores, "kind": "port",
           "sample": "%d evaluations of the same N=%d, D=%d subject (median %.2f s each), NumPy/SciPy oracle of nlogpos_obj in the "
                     "reference's joint-eigenbasis formulation (one N x N eigh per evaluation), value only" % (len(ts), N, M, med)}
    if with_grad:
        t0 = time.perf_counter()
        O.nlogpos_obj(pars, d["Y"], d["x"], **hyper, verbose=True, grad=True)
        tg = time.perf_counter() - t0
        rec["grad"] = {"value": 1.0 / tg, "unit": "evals/s", "sample": "1 value+gradient evaluation (%.2f s), analytic adjoints" % tg}
    return rec


if __name__ == "__main__":
    main()
