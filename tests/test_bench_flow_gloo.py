"""bench.py's N>1 control flow (rank -> subject/chains, warm-up, barrier-bracketed timed steps, max over ranks, the ONE
reduction, rank 0's JSON line) driven by two gloo ranks on CPU with the CPU oracle as evaluator.  The code under test is
bench.run_chains / bench.run_subjects / bench.main themselves -- only the backend object (device, process-group backend,
evaluator) is swapped, exactly the seam bench.py documents.  Reference pattern: Nonseparable_model_mpisim.py:41-43,305-306
(rank -> data file, no communication)."""
import json
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT

N, M = 14, 2


class OracleEvaluator:
    """step(want_grad) -> (out [B, 5], status [B]) like bench.HipChains / bench.HipSubjects, computed by the CPU oracle."""

    def __init__(self, units, hyper, fail=()):
        self.units = units              # list of (x, Y, pars)
        self.hyper = hyper
        self.fail = set(fail)
        self.grads = None
        self.steps_done = 0

    def step(self, want_grad):
        from oracle import nmgp_oracle as O
        outs, st, gs = [], [], []
        for k, (x, Y, p) in enumerate(self.units):
            if k in self.fail:
                outs.append(np.full(5, np.nan))
                st.append(7)
                gs.append(np.zeros_like(p))
                continue
            r = O.nlogpos_obj_SVC(p, Y, x, **self.hyper, verbose=True, grad=want_grad)
            if want_grad:
                r, g = r
                gs.append(g)
            outs.append(np.array(r))
            st.append(0)
        self.grads = np.stack(gs) if want_grad else None
        self.steps_done += 1
        return np.stack(outs), np.array(st, dtype=np.int32)

    def sync(self):
        pass

    def close(self):
        pass


class SeparableOracleEvaluator(OracleEvaluator):
    """bench.HipSeparable's interface on the CPU oracle: out [B, 6] verbose tuples of logpos.nlogpos_obj."""

    def step(self, want_grad):
        from oracle import nmgp_oracle as O
        outs, gs = [], []
        for x, Y, p in self.units:
            r = O.nlogpos_obj(p, Y, x, **self.hyper, verbose=True, grad=want_grad)
            if want_grad:
                r, g = r
                gs.append(g)
            outs.append(np.array(r))
        self.grads = np.stack(gs) if want_grad else None
        self.steps_done += 1
        return np.stack(outs), np.zeros(len(outs), dtype=np.int32)


class GlooOracleBackend:
    device = "cpu"

    def __init__(self, fail_subject=None):
        self.fail_subject = fail_subject
        self.evaluators = []

    def init_dist(self, rank, world):
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def sync(self):
        pass

    def chains(self, d, allp, hv, groups):
        from nonstationary_multivariate_gaussian_process_amd import sim
        ev = OracleEvaluator([(d["x"], d["Y"], p) for p in allp], sim.HYPER_SVC)
        self.evaluators.append(ev)
        return ev

    def subjects(self, subs, pars, hv):
        from nonstationary_multivariate_gaussian_process_amd import sim
        fail = [k for k, d in enumerate(subs) if self.fail_subject is not None and d.get("seed") == self.fail_subject]
        ev = OracleEvaluator([(d["x"], d["Y"], p) for d, p in zip(subs, pars)], sim.HYPER_SVC_MPISIM, fail)
        self.evaluators.append(ev)
        return ev

    def separable(self, d, pars, hv):
        from nonstationary_multivariate_gaussian_process_amd import sim
        ev = SeparableOracleEvaluator([(d["x"], d["Y"], p) for p in pars], sim.HYPER_SEP)
        self.evaluators.append(ev)
        return ev


def _worker(rank, world, port, argv, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import contextlib
    import io
    import bench
    from test_bench_flow_gloo import GlooOracleBackend
    be = GlooOracleBackend()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rec, stats, table = bench.main(argv, backend=be)
    q.put((rank, buf.getvalue(), stats, table, [e.steps_done for e in be.evaluators]))


def _run(argv, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    return got


def bench_peak():
    sys.path.insert(0, ROOT)
    import bench
    return bench.FP64_MATRIX_PEAK_TFLOPS


def test_chain_workload_two_ranks():
    """world = 2, B = 3 chains per rank: one JSON line from rank 0 only, value = steps * world * B / time, the reduction
    covers ALL chains of both ranks (chains_ok == B * world) and every rank holds the same table."""
    B, steps, warm, gsteps = 3, 2, 1, 1
    got = _run(["--gpus", "2", "--steps", str(steps), "--warmup", str(warm), "--N", str(N), "--M", str(M), "--chains",
                str(B), "--grad-steps", str(gsteps), "--cpu-evals", "1", "--cpu-grad-evals", "1"])
    lines0 = [ln for ln in got[0][1].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in got[1][1].splitlines() if ln.startswith("{")]
    rec = json.loads(lines0[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == steps and rec["warmup"] == warm and rec["scaling"] == "weak"
    assert rec["unit"] == "evals/s" and rec["dtype"] == "f64" and rec["vs_baseline"] is None
    assert abs(rec["value"] - steps * 2 * B / (rec["ms_per_step"] * 1e-3 * steps)) < 1e-9 * rec["value"]
    cfg = rec["config"]
    assert cfg["chains_total"] == 2 * B and cfg["chains_ok"] == 2 * B and cfg["chains_failed"] == 0
    assert cfg["chain_table_rows"] == 2 * B
    # every line is self-contained: rank 0 timed the CPU oracle after the timed region, also on this n_gpus = 2 line
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "N=%d" % N in cb["sample"] and cb["grad"]["value"] > 0
    assert cb["reference_formulation"]["value"] > 0
    # flat scalars ahead of the prose: a truncated record still shows them
    keys = list(rec)
    assert keys[:4] == ["metric", "value", "unit", "n_gpus"] and keys.index("value_grad_evals_per_s") < keys.index("config")
    assert keys.index("roofline_frac_end_to_end") < keys.index("config") and keys.index("hmc_samples_per_s") < keys.index("config")
    assert rec["value_grad_evals_per_s"] == rec["grad"]["value"] and rec["cpu_evals_per_s"] == cb["value"]
    assert abs(rec["roofline_frac_end_to_end"] - rec["value"] * float(N * M) ** 3 / 3.0 / 1e12 / 2 / bench_peak()) < 1e-15
    assert rec["hmc_samples_per_s"] is None and "hmc" not in rec          # the sampler object belongs to the N = 1 line
    dr = rec["distributed"]                                # first-contact diagnostics of a multi-GPU run
    assert dr["process_group"] == {"backend": "gloo", "world_size": 2, "rank": 0} and dr["world_size_env"] == 2
    assert len(dr["ms_per_step_by_rank"]) == 2 and dr["ms_per_step_min_rank"] <= dr["ms_per_step_max_rank"]
    assert abs(dr["ms_per_step_max_rank"] - rec["ms_per_step"]) < 1e-9 * rec["ms_per_step"]      # the headline is the slowest rank
    assert rec["grad"]["steps"] == gsteps and rec["grad"]["chains_ok"] == B and rec["grad"]["value"] > 0
    # warm-up + timed + (no profiling pass on this backend) + grad warm-up + grad steps
    assert got[0][4] == [warm + steps + 1 + gsteps] and got[1][4] == [warm + steps + 1 + gsteps]
    stats0, table0 = got[0][2], got[0][3]
    assert np.array_equal(stats0, got[1][2]) and np.array_equal(table0, got[1][3])
    assert table0.shape == (2 * B, 8) and list(table0[:, 0]) == list(range(2 * B)) and np.all(table0[:, 2] == steps)
    # rows equal independent oracle evaluations of (rank's subject, chain's parameters)
    sys.path.insert(0, ROOT)
    import bench
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    tot = 0.0
    for rank in range(2):
        d = sim.simulate_nonseparable(N, M, seed=2222 + rank)
        allp = bench.chain_parameters(sim, d, B)
        for b in range(B):
            ref = np.array(O.nlogpos_obj_SVC(allp[b], d["Y"], d["x"], **sim.HYPER_SVC, verbose=True))
            assert np.array_equal(table0[rank * B + b, 3:8], ref)
            tot += ref[0]
    assert abs(cfg["sum_neglog_all_chains"] - tot) < 1e-9 * abs(tot)


def test_subject_workload_two_ranks():
    """Config-4 pattern at world = 2, 3 subjects per rank: subject s on rank s mod world with seed s, table ordered by
    subject id on every rank."""
    got = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--N", str(N), "--M", str(M), "--workload", "subjects",
                "--subjects-per-gpu", "3"])
    rec = json.loads([ln for ln in got[0][1].splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["config"]["subjects_total"] == 6 and rec["config"]["subjects_ok"] == 6
    assert rec["config"]["subject_table_rows"] == 6
    assert rec["distributed"]["process_group"]["world_size"] == 2 and len(rec["distributed"]["ms_per_step_by_rank"]) == 2
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_evals_per_s"] == rec["cpu_baseline"]["value"]
    assert rec["roofline"]["traffic"] is None and "no committed PMC" in rec["roofline"]["traffic_note"]
    assert rec["scaling"] == "weak" and "subjects-total" in rec["scaling_note"] and rec["roofline_frac_end_to_end"] > 0
    table = got[0][3]
    assert np.array_equal(table, got[1][3]) and list(table[:, 0]) == list(range(6))
    sys.path.insert(0, ROOT)
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    for s in range(6):
        d = sim.simulate_nonseparable(N, M, seed=s)
        ref = np.array(O.nlogpos_obj_SVC(sim.perturb(d["pars_true"], 0.05, 0.7), d["Y"], d["x"], **sim.HYPER_SVC_MPISIM,
                                         verbose=True))
        assert np.array_equal(table[s, 3:8], ref)


def test_a_fixed_subject_set_is_labelled_strong_scaling():
    """Config 4 as worded: a FIXED number of subjects sharded over the GPUs (--subjects-total) -- the per-GPU batch shrinks with
    N, so the line says "strong"; 5 subjects over 2 ranks = 3 + 2 (s mod world)."""
    got = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--N", str(N), "--M", str(M), "--workload", "subjects",
                "--subjects-total", "5", "--cpu-evals", "1"])
    rec = json.loads([ln for ln in got[0][1].splitlines() if ln.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["config"]["subjects_total"] == 5 and rec["config"]["subjects_ok"] == 5
    assert "5 independent subjects (3 on rank 0" in rec["config"]["workload"]
    table = got[0][3]
    assert list(table[:, 0]) == list(range(5)) and np.array_equal(table, got[1][3])
    assert got[0][4] == [1] and got[1][4] == [1]
    assert "all_subjects_on_one_gpu" not in rec          # a single-GPU side measurement only


def test_separable_workload_two_ranks():
    """BASELINE config 5's workload through the same control flow: B chains of the separable model per rank, rows = the oracle's
    verbose tuples, roofline on D N^3 / 3 flop per evaluation, cpu_baseline from the oracle's nlogpos_obj."""
    B, steps = 2, 1
    got = _run(["--gpus", "2", "--steps", str(steps), "--warmup", "0", "--N", str(N), "--M", str(M), "--workload", "separable",
                "--chains", str(B), "--grad-steps", "1"])
    lines0 = [ln for ln in got[0][1].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in got[1][1].splitlines() if ln.startswith("{")]
    rec = json.loads(lines0[0])
    assert rec["metric"] == "log-posterior evals/sec (N=%d, D=%d separable GP)" % (N, M) and rec["n_gpus"] == 2
    assert "BASELINE config 5" in rec["config"]["workload"] and rec["config"]["chains_ok"] == 2 * B
    assert abs(rec["value"] - steps * 2 * B / (rec["ms_per_step"] * 1e-3 * steps)) < 1e-9 * rec["value"]
    assert abs(rec["roofline_frac_end_to_end"] - rec["value"] * M * float(N) ** 3 / 3.0 / 1e12 / 2 / bench_peak()) < 1e-15
    assert rec["grad"]["value"] > 0 and rec["value_grad_evals_per_s"] == rec["grad"]["value"]
    assert abs(rec["grad"]["roofline"]["frac"] - rec["grad"]["value"] * M * float(N) ** 3 / 1e12 / 2 / bench_peak()) < 1e-15
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["grad"]["value"] > 0 and "nlogpos_obj" in cb["sample"]
    for k in ("roofline", "distributed", "scaling", "dtype", "data", "vs_baseline", "higher_is_better"):
        assert k in rec
    table = got[0][3]
    sys.path.insert(0, ROOT)
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    for rank in range(2):
        d = sim.simulate_separable(N, M, 8 + rank)
        for b in range(B):
            ref = np.array(O.nlogpos_obj(sim.perturb(d["pars_true"], 0.05, 0.4 + 0.1 * b), d["Y"], d["x"], **sim.HYPER_SEP, verbose=True))
            assert np.array_equal(table[rank * B + b, 3:8], ref[:5])


def test_measured_traffic_lookup_is_keyed_by_workload_and_pinned_to_the_kernel_source(tmp_path, monkeypatch):
    """roofline.traffic comes from committed PMC summaries (profiles/traffic.json): one entry per (N, M, batch, value / value+gradient,
    workload), reported only while csrc/nmgp_chol.hip still has the SHA-256 it was measured on -- otherwise null with the reason."""
    sys.path.insert(0, ROOT)
    import hashlib
    import bench
    from nonstationary_multivariate_gaussian_process_amd import build
    sha = hashlib.sha256(open(bench.CHOL_SOURCE, "rb").read()).hexdigest()
    tree = build.tree_id()
    syrk, whole = "k_syrk_lower launches", "every kernel of the evaluation"
    doc = {"entries": [
        {"N": 2048, "M": 3, "chains": 128, "grad": False, "workload": "chain", "bytes_per_step": 1.0, "source": "a", "chol_sha256": sha,
         "scope": syrk, "tree_id": "f" * 64},        # a k_syrk_lower figure follows csrc/nmgp_chol.hip only
        {"N": 2048, "M": 3, "chains": 128, "grad": True, "workload": "chain", "bytes_per_step": 2.0, "source": "b", "chol_sha256": sha,
         "scope": whole, "tree_id": tree},
        {"N": 1024, "M": 3, "chains": 8, "grad": False, "workload": "subjects", "bytes_per_step": 3.0, "source": "c", "chol_sha256": sha,
         "scope": whole, "tree_id": tree},
        {"N": 1024, "M": 3, "chains": 64, "grad": False, "workload": "subjects", "bytes_per_step": 4.0, "source": "d",
         "chol_sha256": sha, "scope": whole, "tree_id": "0" * 64},      # whole-evaluation figure of ANOTHER build: stale
        {"N": 512, "M": 3, "chains": 4, "grad": False, "workload": "chain", "bytes_per_step": 5.0, "source": "e",
         "chol_sha256": "0" * 64, "scope": syrk, "tree_id": tree}]}
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    assert bench.measured_traffic(2048, 3, 128)[0] == 1.0
    assert bench.measured_traffic(2048, 3, 128, True)[0] == 2.0
    assert bench.measured_traffic(1024, 3, 8, False, "subjects")[0] == 3.0
    assert bench.measured_traffic(1024, 3, 8, False, "subjects")[2] == "every kernel of the evaluation"      # -> roofline.traffic_scope
    v, why, _ = bench.measured_traffic(1024, 3, 8, False, "chain")
    assert v is None and "no committed PMC measurement" in why
    v, why, scope = bench.measured_traffic(1024, 3, 64, False, "subjects")
    assert v is None and "stale" in why and "build" in why
    v, why, scope = bench.measured_traffic(512, 3, 4)
    assert v is None and "stale" in why and "nmgp_chol.hip" in why
    # a schedule-changing switch in the environment: the committed passes ran the default schedule
    monkeypatch.setenv("NMGP_TRTRI", "1")
    v, why, _ = bench.measured_traffic(2048, 3, 128, True)
    assert v is None and "NMGP_TRTRI" in why
    monkeypatch.delenv("NMGP_TRTRI")
    monkeypatch.setenv("NMGP_ROUND", "r05")             # (not a schedule switch)
    assert bench.measured_traffic(2048, 3, 128, True)[0] == 2.0
    # the committed file itself: entries for the headline, its value+gradient step and config 4's per-GPU shape
    monkeypatch.undo()
    doc = json.load(open(bench.TRAFFIC_FILE))
    keys = {(e["N"], e["M"], e["chains"], bool(e.get("grad", False)), e.get("workload", "chain")) for e in doc["entries"]}
    assert {(2048, 3, 128, False, "chain"), (2048, 3, 128, True, "chain"), (1024, 3, 8, False, "subjects")} <= keys


def test_single_rank_dist_selftest_runs_the_collectives():
    """--dist-selftest: ONE rank initialises the process group and the barrier, max-over-ranks, per-rank gather and the final
    reduction go through torch.distributed (gloo here; the GPU twin of this test does it with RCCL next to the library's streams)."""
    B, steps = 2, 1
    got = _run(["--gpus", "1", "--steps", str(steps), "--warmup", "0", "--N", str(N), "--M", str(M), "--chains", str(B),
                "--grad-steps", "0", "--no-cpu-baseline", "--dist-selftest"], world=1)
    rec = json.loads([ln for ln in got[0][1].splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["distributed"]["process_group"] == {"backend": "gloo", "world_size": 1, "rank": 0}
    assert rec["config"]["chains_ok"] == B and got[0][3].shape == (B, 8)
