"""Every selectable kernel variant of the blocked factorisation (environment switches, read once per process) must give
the same objective and gradient as the default configuration: each variant runs in its own process on a golden case
whose size exercises full tiles, half-width tiles, the ragged last block and the right-hand-side row."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr, vec_relerr

pytestmark = pytest.mark.gpu

SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from nonstationary_multivariate_gaussian_process_amd import _lib, sim
d = sim.simulate_nonseparable(200, 3, seed=7)             # n = 600: 9 blocks of 64 + one of 24
pars = sim.perturb(d["pars_true"], 0.05, 0.3)
hv = [sim.HYPER_SVC[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")]
ctx = _lib.Context(0)
ctx.set_data(d["x"], d["Y"])
out, grad = ctx.logpos_svc(pars, hv, prior=True, want_grad=True)
B = 5
ctx.svc_batch_alloc(B)
allp = np.stack([sim.perturb(d["pars_true"], 0.05, 0.3 + 0.1 * b) for b in range(B)])
ctx.svc_batch_set_pars(allp)
ctx.svc_batch_eval(hv, True, want_grad=False)
bout, status = ctx.svc_batch_fetch()
# a second subject whose size is a whole number of 64-column blocks (n = 576 = one 512-wide panel + one block): the shape the
# fused panel steps take; batched value + gradient so that the L^-T rows ride along
d2 = sim.simulate_nonseparable(192, 3, seed=8)
ctx.set_data(d2["x"], d2["Y"])
out2, grad2 = ctx.logpos_svc(sim.perturb(d2["pars_true"], 0.05, 0.2), hv, prior=True, want_grad=True)
ctx.svc_batch_alloc(3)
allp2 = np.stack([sim.perturb(d2["pars_true"], 0.05, 0.2 + 0.1 * b) for b in range(3)])
ctx.svc_batch_set_pars(allp2)
ctx.svc_batch_eval(hv, True, want_grad=True)
bout2, status2 = ctx.svc_batch_fetch()
bgrad2 = ctx.svc_batch_fetch_grad()
# four chains, value + gradient: the smallest batch that takes the throughput schedule once NMGP_CHOL_FUSED_MAX_BATCH=0 rules out
# the fused steps -- 128-column leaves (k_panel_step<1>, <2>) with the L^-T rows riding along
ctx.svc_batch_alloc(4)
allp4 = np.stack([sim.perturb(d2["pars_true"], 0.05, 0.2 + 0.1 * b) for b in range(4)])
ctx.svc_batch_set_pars(allp4)
ctx.svc_batch_eval(hv, True, want_grad=True)
bout4, status4 = ctx.svc_batch_fetch()
bgrad4 = ctx.svc_batch_fetch_grad()
# a third subject with more than two 512-wide panels (n = 1200 = 512 + 512 + 176): the look-ahead schedule, its near update on
# 64x64 tiles with the next panel's first block factored in the same launch, and a ragged last panel that takes neither
d3 = sim.simulate_nonseparable(400, 3, seed=9)
ctx.set_data(d3["x"], d3["Y"])
out3, grad3 = ctx.logpos_svc(sim.perturb(d3["pars_true"], 0.05, 0.2), hv, prior=True, want_grad=True)
print(json.dumps({"out3": list(map(float, out3)), "grad3": list(map(float, grad3)), "out": list(map(float, out)), "grad": list(map(float, grad)), "batch": bout.tolist(),
                  "status": status.tolist() + status2.tolist() + status4.tolist(), "batch4": bout4.tolist(), "bgrad4": bgrad4.tolist(), "out2": list(map(float, out2)),
                  "grad2": list(map(float, grad2)), "batch2": bout2.tolist(), "bgrad2": bgrad2.tolist()}))
"""

VARIANTS = [{}, {"NMGP_TRSM": "valu"}, {"NMGP_TRSM": "f"}, {"NMGP_POTF2": "valu"}, {"NMGP_TRSM": "valu", "NMGP_POTF2": "valu"},
            {"NMGP_SYRK_YROW": "0"},
            {"NMGP_CHOL_PANEL": "fused"}, {"NMGP_CHOL_PANEL": "rec"}, {"NMGP_CHOL_PANEL": "rl"},
            {"NMGP_CHOL_PANEL": "fused", "NMGP_CHOL_NB1": "128"},
            {"NMGP_CHOL_LOOKAHEAD": "1", "NMGP_CHOL_NB1": "128"}, {"NMGP_CHOL_NB1": "256"},
            {"NMGP_CHOL_LOOKAHEAD": "0"}, {"NMGP_LOOKAHEAD_CUS": "0", "NMGP_CHOL_NB1": "128"}, {"NMGP_PRIOR_OVERLAP": "0"},
            {"NMGP_CHOL_FUSED_MAX_BATCH": "0"}, {"NMGP_PRIOR_SOLVE": "rocblas"},
            {"NMGP_CHOL_FUSE_POTF2": "0"}, {"NMGP_SYRK_SMALL_MAX": "0"}, {"NMGP_SYRK_SMALL_MAX": "100000"},
            {"NMGP_POISON": "1"}, {"NMGP_POISON": "1", "NMGP_CHOL_PANEL": "fused"},
            # fused steps under a recursive split (round 3), with the L^-T rows that enter panel by panel; substitution prior solves
            {"NMGP_CHOL_FUSED_BASE": "128"}, {"NMGP_CHOL_FUSED_BASE": "256", "NMGP_CHOL_PANEL": "fused"},
            {"NMGP_CHOL_FUSED_BASE": "128", "NMGP_CHOL_PANEL": "fused", "NMGP_POISON": "1"}, {"NMGP_PRIOR_SOLVE": "trsv"},
            {"NMGP_SYRK_TRI_ORDER": "strips"},
            # the throughput schedule on the snippet's small batches: leaf launches (default), with poisoned buffers, and the
            # five-launch form they replace
            {"NMGP_CHOL_FUSED_MAX_BATCH": "0", "NMGP_POISON": "1"}, {"NMGP_CHOL_FUSED_MAX_BATCH": "0", "NMGP_CHOL_LEAF": "0"},
            # ... and the other register budgets of the two leaf kernels (defaults: first step 4 waves per SIMD, second step 6)
            {"NMGP_CHOL_FUSED_MAX_BATCH": "0", "NMGP_LEAF1_OCC": "6", "NMGP_LEAF2_OCC": "4"}]


def run_variant(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], capture_output=True, text=True, timeout=600,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, (env_extra, out.stderr[-2000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_kernel_variants_agree_with_the_default_configuration():
    ref = run_variant(VARIANTS[0])
    assert all(s == 0 for s in ref["status"])
    for env_extra in VARIANTS[1:]:
        r = run_variant(env_extra)
        assert all(s == 0 for s in r["status"]), env_extra
        # likelihood term: rounding-level agreement; totals: the conditioning noise of the prior terms is shared (cached
        # prior factors always use the substitution kernels)
        assert relerr(r["out"][1], ref["out"][1]) < 1e-11, (env_extra, r["out"], ref["out"])
        assert relerr(np.array(r["out"]), np.array(ref["out"])) < 1e-7, env_extra
        assert vec_relerr(np.array(r["grad"]), np.array(ref["grad"])) < 1e-7, env_extra
        assert relerr(np.array(r["batch"]), np.array(ref["batch"])) < 1e-7, env_extra
        assert relerr(r["out2"][1], ref["out2"][1]) < 1e-11, (env_extra, r["out2"], ref["out2"])
        assert relerr(np.array(r["out2"]), np.array(ref["out2"])) < 1e-7, env_extra
        assert vec_relerr(np.array(r["grad2"]), np.array(ref["grad2"])) < 1e-7, env_extra
        assert relerr(np.array(r["batch2"]), np.array(ref["batch2"])) < 1e-7, env_extra
        assert vec_relerr(np.array(r["bgrad2"]), np.array(ref["bgrad2"])) < 1e-7, env_extra
        assert relerr(np.array(r["batch4"]), np.array(ref["batch4"])) < 1e-7, env_extra
        assert vec_relerr(np.array(r["bgrad4"]), np.array(ref["bgrad4"])) < 1e-7, env_extra
        assert relerr(r["out3"][1], ref["out3"][1]) < 1e-11, (env_extra, r["out3"], ref["out3"])
        assert relerr(np.array(r["out3"]), np.array(ref["out3"])) < 1e-7, env_extra
        assert vec_relerr(np.array(r["grad3"]), np.array(ref["grad3"])) < 1e-7, env_extra


def test_parity_suite_passes_with_nan_poisoned_device_buffers():
    """NMGP_POISON=1 fills every fresh device buffer and every scratch hand-out with NaNs: a kernel or library call that
    reads memory nobody wrote (this caught a beta = 0 dsymm into stale scratch and a zero-weighted contraction over
    unwritten LDS slots) then fails the parity suite deterministically instead of depending on allocator history."""
    env = dict(os.environ)
    env["NMGP_POISON"] = "1"
    env["NMGP_ROUND"] = "poison"              # its achieved-error table must not overwrite the main run's
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu",
                          "-x", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]
    # ... and the sampler's metric kernels (round 5): coordinate changes, trajectories against the dense-mass path, batched = single
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hmc_metric.py"), "-q", "-m", "gpu", "-x",
                          "-p", "no:cacheprovider", "-k", "prior_apply or trajectories or bits or multi_subject"],
                         capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("variant", ["1", "3"])
def test_separable_batch_under_the_selectable_block_assembly_kernels(variant):
    """NMGP_SEP_BLOCKS: the batched separable evaluation's block-assembly kernel in its other forms (64 x 64 tiles in grid order /
    in the XCD-aware order; the default is 128 x 32 tiles with 16-byte stores) against the same goldens and single-chain evaluations."""
    env = dict(os.environ)
    env["NMGP_SEP_BLOCKS"] = variant
    env["NMGP_ROUND"] = "variants"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                          "-p", "no:cacheprovider", "-k", "separable_chains_batched"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:]


@pytest.mark.parametrize("env_extra", [
    {"NMGP_SYRK_SMALL_MAX": "100000", "NMGP_CHOL_NB1": "128"},      # every K >= 128 update on 64x64 tiles, look-ahead from n > 256
    {"NMGP_SYRK_SMALL_MAX": "100000", "NMGP_CHOL_NB1": "192", "NMGP_CHOL_PANEL": "rl"},   # panels that are no multiple of 128
    {"NMGP_CHOL_NB1": "64"},                                        # one step per panel: every update is a trailing update
])
def test_factorisation_suite_under_small_tile_and_narrow_panel_settings(env_extra):
    """The LAPACK comparisons of the custom factorisation (ragged sizes 1 .. 2112, right-hand-side row, indefinite inputs)
    with the 64x64-tile update kernel forced onto every launch it can take, and with panel widths that move the panel
    boundaries -- and with them the shapes the near / far updates and the fused first block see -- to other places."""
    env = dict(os.environ)
    env.update(env_extra)
    env["NMGP_ROUND"] = "variants"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu",
                          "-x", "-p", "no:cacheprovider", "-k", "custom_cholesky"], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (env_extra, out.stdout[-3000:])


def test_bench_dist_selftest_runs_rccl_next_to_the_library_streams():
    """`bench.py --dist-selftest` on one GPU: the process group is initialised with the nccl (= RCCL) backend and the barrier, the
    max-over-ranks all-reduce, the per-rank all-gather and the final reduction run through it in the same process as the library's
    three streams -- the part of the multi-GPU path (HipBackend.init_dist + collectives on CUDA tensors) a single-GPU box can execute."""
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dist-selftest", "--chains", "3", "--N", "96", "--steps", "2",
                          "--warmup", "1", "--grad-steps", "1", "--hmc-samples", "0", "--no-cpu-baseline"], capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["distributed"]["process_group"] == {"backend": "nccl", "world_size": 1, "rank": 0}
    assert rec["config"]["chains_ok"] == 3 and rec["config"]["chain_table_rows"] == 3 and rec["value"] > 0
    assert rec["grad"]["chains_ok"] == 3


TRTRI_SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from nonstationary_multivariate_gaussian_process_amd import _lib, sim
from oracle import nmgp_oracle as O          # checker
hv = [sim.HYPER_SVC[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")]
res = {}
ctx = _lib.Context(0)
for (N, M, B) in ((128, 2, 3), (128, 3, 1), (256, 3, 5), (512, 3, 2)):       # n = 256, 384, 768, 1536: 2, 3, 6, 12 leaf blocks
    d = sim.simulate_nonseparable(N, M, seed=11 + N + M)
    ctx.set_data(d["x"], d["Y"])
    p0 = sim.perturb(d["pars_true"], 0.05, 0.2)
    out, grad = ctx.logpos_svc(p0, hv, prior=True, want_grad=True)
    ref, gref = O.nlogpos_obj_SVC(p0, d["Y"], d["x"], **sim.HYPER_SVC, verbose=True, grad=True)
    key = "N%%d_M%%d" %% (N, M)
    res[key] = {"out": list(map(float, out)), "grad": list(map(float, grad)),
                "oracle_rel": float(abs(out[0] - ref[0]) / abs(ref[0])),
                "oracle_grad_rel": float(np.linalg.norm(grad - gref) / np.linalg.norm(gref))}
    if B > 1:
        ctx.svc_batch_alloc(B)
        allp = np.stack([sim.perturb(d["pars_true"], 0.05, 0.2 + 0.1 * b) for b in range(B)])
        ctx.svc_batch_set_pars(allp)
        ctx.svc_batch_eval(hv, True, want_grad=True)
        bout, status = ctx.svc_batch_fetch()
        res[key].update(batch=bout.tolist(), bgrad=ctx.svc_batch_fetch_grad().tolist(), status=status.tolist())
print(json.dumps(res))
"""


def test_blocked_triangular_inversion_agrees_with_the_riding_rows_and_the_oracle():
    """NMGP_TRTRI=1: gradient evaluations whose n is a multiple of 128 build X = L^-T AFTER the factorisation (nmgp_trtri.hip: leaf
    blocks by substitution, then products with explicit inverses of the diagonal blocks) instead of the default (identity rows
    riding through the factorisation).  Both must give the same objective and gradient -- sizes with 2, 3, 6 and 12 leaf blocks, i.e.
    power-of-two levels only, the left-to-right top combine only, and both --, with NaN-poisoned buffers too, and match the oracle."""
    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        out = subprocess.run([sys.executable, "-c", TRTRI_SNIPPET % {"root": ROOT}], capture_output=True, text=True, timeout=900,
                             env=env, cwd=ROOT)
        assert out.returncode == 0, (env_extra, out.stderr[-2000:])
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    ref = run({"NMGP_TRTRI": "0"})
    for env_extra in ({"NMGP_TRTRI": "1"}, {"NMGP_TRTRI": "1", "NMGP_POISON": "1"}, {"NMGP_TRTRI": "1", "NMGP_CHOL_FUSED_MAX_BATCH": "0"},
                      {"NMGP_TRTRI": "1", "NMGP_TRTRI_ORDER": "lockstep"}, {"NMGP_TRTRI": "1", "NMGP_TRTRI_ORDER": "rows"}):
        r = run(env_extra)
        for key, a in r.items():
            b = ref[key]
            assert a["oracle_rel"] < 1e-6 and a["oracle_grad_rel"] < 1e-5, (env_extra, key, a["oracle_rel"], a["oracle_grad_rel"])
            assert relerr(a["out"][1], b["out"][1]) < 1e-11, (env_extra, key)
            assert vec_relerr(np.array(a["grad"]), np.array(b["grad"])) < 1e-7, (env_extra, key)
            if "batch" in a:
                assert all(s == 0 for s in a["status"])
                assert relerr(np.array(a["batch"]), np.array(b["batch"])) < 1e-7, (env_extra, key)
                assert vec_relerr(np.array(a["bgrad"]), np.array(b["bgrad"])) < 1e-7, (env_extra, key)


def test_bench_launches_two_ranks_by_itself_and_rehearses_the_multi_process_path_on_one_gpu():
    """`python bench.py --gpus 2 --rehearse-on-one-gpu` with no torch.distributed environment: bench.py starts its own two ranks (a
    child `python -m torch.distributed.run`, before anything touches the GPU), each rank loads the library, creates its context and
    streams on the box's one GPU and runs the real evaluations; barrier, max-over-ranks, gathers and the final reduction go through a
    gloo process group (RCCL refuses two ranks on one device).  What a one-GPU box can execute of the N > 1 path, as one command."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--chains", "3", "--N", "96",
                          "--steps", "2", "--warmup", "1", "--grad-steps", "1", "--hmc-samples", "1", "--hmc-all-ranks", "--hmc-mass", "identity",
                          "--cpu-evals", "1", "--cpu-grad-evals", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    dr = rec["distributed"]
    assert rec["n_gpus"] == 2 and rec["rehearsal"] is True and dr["self_launched"] is True
    assert dr["process_group"] == {"backend": "gloo", "world_size": 2, "rank": 0}
    assert [r["rank"] for r in dr["ranks"]] == [0, 1] and [r["local_rank"] for r in dr["ranks"]] == [0, 1]
    assert len({r["pid"] for r in dr["ranks"]}) == 2 and all(r["device_ordinal"] == 0 for r in dr["ranks"])
    assert len({r["library_build_id"] for r in dr["ranks"]}) == 1 and dr["distinct_gpus"] == 1
    assert rec["config"]["chains_total"] == 6 and rec["config"]["chains_ok"] == 6 and rec["grad"]["chains_ok"] == 3
    assert len(rec["hmc"]["by_rank"]) == 2 and rec["hmc"]["by_rank"][1]["rank"] == 1      # (--hmc-all-ranks: off by default when N > 1)
    assert rec["cpu_baseline"]["value"] > 0 and rec["hmc_samples_per_s"] == rec["hmc"]["samples_per_s"]      # a self-contained N = 2 line
