"""bench.py contract: one JSON line with the required keys, `roofline` and `cpu_baseline` objects (GPU), and the CPU
baseline helper on its own (CPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def test_cpu_baseline_helper_runs_on_cpu():
    sys.path.insert(0, ROOT)
    import bench
    from nonstationary_multivariate_gaussian_process_amd import sim
    d = sim.simulate_nonseparable(48, 2, seed=1)
    rec = bench.cpu_baseline(d, d["pars_true"], sim.HYPER_SVC, 1, False)
    assert rec["kind"] == "port" and rec["unit"] == "evals/s" and rec["value"] > 0 and rec["cores"] >= 1
    assert "sample" in rec and "N=48" in rec["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--chains", "3"], ["--chains", "1", "--grad"],
                                   ["--workload", "subjects", "--subjects-per-gpu", "3"]])
def test_bench_prints_one_valid_json_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--N", "96", "--M", "3", "--steps", "2", "--warmup", "1",
           "--cpu-evals", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    if "subjects" in extra:
        assert rec["value"] > 0 and rec["config"]["subjects_ok"] == 3 and rec["scaling"] == "weak"
        return
    for k in REQUIRED:
        assert k in rec, k
    assert rec["unit"] == "evals/s" and rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["dtype"] == "f64"
    assert rec["higher_is_better"] is True and rec["vs_baseline"] is None and rec["data"] == "synthetic"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    rl = rec["roofline"]
    assert rl["bound"] in ("hbm", "mfma") and rl["unit"] == "TFLOP/s" and rl["peak"] > 0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12 and "traffic" in rl
    cb = rec["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["reference_formulation"]["value"] > 0 and "inverse + logdet" in cb["reference_formulation"]["sample"]
    B = int(extra[extra.index("--chains") + 1])
    assert rec["config"]["chains_total"] == B and rec["config"]["chains_ok"] == B      # every chain is reduced, not chain 0 only
    if "--grad" not in extra:
        gr = rec["grad"]
        assert gr["value"] > 0 and gr["chains_ok"] == B and gr["roofline"]["bound"] == "mfma" and gr["grad_norm_chain0"] > 0
        assert abs(gr["roofline"]["frac"] - gr["roofline"]["achieved"] / gr["roofline"]["peak"]) < 1e-12
        if B > 1:
            assert rec["hmc"]["samples_per_s"] > 0 and 0.0 <= rec["hmc"]["accept_rate_mean"] <= 1.0
    else:
        assert "grad" not in rec and "gradients" in rec["config"]["host_reads_per_step"]
    assert rl["traffic"] is None and rl["traffic_note"]          # no PMC measurement exists for this toy size
    assert np.isfinite(rec["value"]) and rec["value"] > 0


def test_profile_tools_replay_the_launch_schedule_of_the_throughput_path():
    """tools/syrk_classes.py attaches (rows, columns, K) to the k_syrk_lower launches of a trace by replaying the host
    schedule: recursive halving of 2048-wide outer panels down to the 128-column leaves (two k_panel_step launches, no
    update launch) -- or down to 64 columns when NMGP_CHOL_LEAF=0.  Flop of the replayed launches + the leaves' K = 64 share
    must add up to the factorisation's update flop."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import syrk_classes
    n, nb1 = 6144, 2048
    old = syrk_classes.LEAF
    try:
        syrk_classes.LEAF = True
        leaf = syrk_classes.schedule(n, nb1)
        syrk_classes.LEAF = False
        full = syrk_classes.schedule(n, nb1)
    finally:
        syrk_classes.LEAF = old
    assert len(leaf) == 47 and len(full) == 95           # the launch counts in profiles/r03_batched128_last_eval.txt / r02
    assert sorted(set(t[2] for t in full)) == [64, 128, 256, 512, 1024, 2048]
    assert sorted(set(t[2] for t in leaf)) == [128, 256, 512, 1024, 2048]
    assert [t for t in full if t[2] != 64] == leaf       # the leaves only remove the K = 64 launches
    assert all(c1 - c0 == k for _, _, k, c0, c1 in full)  # K = width of the panel [c0, c1) being applied

    def flop(sched):
        return sum(2.0 * k * (nc * m - 0.5 * nc * (nc - 1)) for m, nc, k, _, _ in sched)
    # all update launches together: n^3/3 minus the diagonal blocks and panel solves (98.5 % at n = 6144, one extra row)
    assert 0.98 < flop(full) / (n ** 3 / 3.0) < 0.99
    assert 0.96 < flop(leaf) / (n ** 3 / 3.0) < flop(full) / (n ** 3 / 3.0)
    # value evaluation: executed flop == nominal flop, launch by launch
    assert all(syrk_classes.launch_flop(m, nc, k, c0, c1, n, 1, 0) == 2.0 * k * (nc * m - 0.5 * nc * (nc - 1)) for m, nc, k, c0, c1 in leaf)
    # value+gradient evaluation (n rows of L^-T ride below): the tiles of those rows skip their leading zero k-panels, so the
    # executed flop of a launch is below 2 K x elements (what round 3's table counted: 88.7 / 93.6 "TFLOP/s" at K = 2048) and the
    # total lies between the structural minimum 2 n^3/3 (less the leaves' share) and the nominal count
    g = syrk_classes.schedule(n, nb1, 2, n)
    ex = sum(syrk_classes.launch_flop(m, nc, k, c0, c1, n, 2, n) for m, nc, k, c0, c1 in g)
    nom = flop(g)
    assert len(g) == 47 and ex < nom and 0.95 * (2 * n ** 3 / 3.0) < ex < 1.03 * (2 * n ** 3 / 3.0) and nom > 1.12 * (2 * n ** 3 / 3.0)
    big = [t for t in g if t[2] == 2048]
    assert all(syrk_classes.launch_flop(*t, n, 2, n) < 0.9 * 2.0 * t[2] * (t[1] * t[0] - 0.5 * t[1] * (t[1] - 1)) for t in big)


def test_committed_class_tables_never_exceed_the_matrix_peak():
    """Every per-class TFLOP/s figure of the committed round-4 tables (tools/syrk_classes.py output under profiles/) is at most the FP64
    matrix peak: a figure above it means the timed kernel is not doing the counted work."""
    import glob
    import re
    sys.path.insert(0, ROOT)
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_*syrk_classes.txt")))
    for f in files:
        for ln in open(f):
            for v in re.findall(r"([0-9.]+) TF/s", ln):
                assert float(v) <= bench.FP64_MATRIX_PEAK_TFLOPS, (f, ln)
