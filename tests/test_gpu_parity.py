"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (ctypes) and through
the Python `Utility` mirror, against (1) the golden vectors generated from the reference and (2) the CPU oracle on
seeded inputs.  Tolerances: log posterior 1e-6 relative (north star), likelihood term 1e-9, gradients 1e-5
(||dg||/||g||), covariance entries 1e-13."""
import numpy as np
import pytest

from conftest import (SEP_KEYS, STA_KEYS, SVC_KEYS, golden, golden_names, hyper_dict, prior_component_err_on_the_logdet_scale, record_parity, relerr,
                      vec_relerr)

pytestmark = pytest.mark.gpu

VAL_TOL = 1e-6
LIK_TOL = 1e-9
GRAD_TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from nonstationary_multivariate_gaussian_process_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def test_native_library_is_what_runs(ctx):
    """The extension in-tree is loaded and the device is real (no fallback exists to hide behind)."""
    from nonstationary_multivariate_gaussian_process_amd import _lib
    assert _lib.load().nmgp_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libnmgp_hip.so" in maps


@pytest.mark.parametrize("name", golden_names("svc_"))
def test_svc_against_reference_golden(ctx, name):
    g = golden(name)
    ctx.set_data(g["x"], g["Y"])
    out, grad = ctx.logpos_svc(g["pars"], g["hyper"], prior=bool(g["prior"]), want_grad=True)
    record_parity(name, neglog=(relerr(out[0], g["out"][0]), VAL_TOL), loglik=(relerr(out[1], g["out"][1]), LIK_TOL),
                  priors=(relerr(out[2:], g["out"][2:]), VAL_TOL), grad=(vec_relerr(grad, g["grad"]), GRAD_TOL))
    assert relerr(out[0], g["out"][0]) < VAL_TOL, (out, g["out"])
    assert relerr(out[1], g["out"][1]) < LIK_TOL
    assert relerr(out[2:], g["out"][2:]) < VAL_TOL
    assert vec_relerr(grad, g["grad"]) < GRAD_TOL
    # value-only evaluation gives the same numbers
    out2, _ = ctx.logpos_svc(g["pars"], g["hyper"], prior=bool(g["prior"]), want_grad=False)
    assert np.array_equal(out, out2)
    if "Sigma" in g:
        S = ctx.svc_covariance(g["pars"])
        assert np.allclose(S, g["Sigma"], rtol=1e-13, atol=1e-15)
        assert np.array_equal(S, S.T)


@pytest.mark.parametrize("N,M,seed", [(1, 1, 0), (2, 2, 1), (63, 3, 2), (65, 2, 3), (130, 4, 4), (200, 5, 5), (40, 6, 6),
                                      (33, 7, 7), (20, 8, 8)])
def test_svc_against_oracle_ragged_sizes(ctx, N, M, seed):
    """Sizes that are not tile multiples, single location / single output, every supported M."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    d = sim.simulate_nonseparable(N, M, seed)
    pars = sim.perturb(d["pars_true"], 0.05, 0.1 * seed)
    for hyper in (sim.HYPER_SVC, sim.HYPER_SVC_DIST):
        hv = [hyper[k] for k in SVC_KEYS]
        ctx.set_data(d["x"], d["Y"])
        out, grad = ctx.logpos_svc(pars, hv, prior=True, want_grad=True)
        ref, gref = O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **hyper, verbose=True, grad=True)
        assert relerr(out[0], ref[0]) < VAL_TOL, (out, ref)
        assert relerr(out[1], ref[1]) < LIK_TOL
        assert vec_relerr(grad, gref) < GRAD_TOL
    tl, uL, tse = O.vec2pars_SVC(pars, N, M)
    assert np.allclose(ctx.svc_covariance(pars), O.svc_covariance(tl, uL, tse, d["x"], M), rtol=1e-13, atol=1e-15)


def test_svc_prior_flag_and_resident_api(ctx):
    g = golden("svc_rngfree_N64_M3")
    ctx.set_data(g["x"], g["Y"])
    full, _ = ctx.logpos_svc(g["pars"], g["hyper"], prior=True)
    lik, _ = ctx.logpos_svc(g["pars"], g["hyper"], prior=False)
    assert lik[0] == -lik[1]                       # Prior=False: res is the likelihood alone (logpos.py:355,359)
    assert np.array_equal(full[1:], lik[1:])       # the verbose components are still reported
    tse = g["pars"][-1]
    assert relerr(full[0], -(full[1] + full[2] + full[3] + full[4] + tse)) < 1e-14
    # resident form: parameters stay in HBM, repeated evaluations are bit-identical
    ctx.svc_set_pars(g["pars"])
    ctx.svc_eval_resident(g["hyper"], True, True)
    o1, g1 = ctx.svc_fetch(True)
    ctx.svc_eval_resident(g["hyper"], True, True)
    o2, g2 = ctx.svc_fetch(True)
    assert np.array_equal(o1, o2) and np.array_equal(g1, g2) and np.array_equal(o1, full)


def test_svc_error_behaviour(ctx):
    from nonstationary_multivariate_gaussian_process_amd import _lib
    g = golden("svc_rngfree_N8_M3")
    ctx.set_data(g["x"], g["Y"])
    with pytest.raises(_lib.NmgpError):
        ctx.logpos_svc(g["pars"][:-1], g["hyper"])          # wrong length
    bad = g["pars"].copy()
    bad[3] = np.nan
    with pytest.raises(_lib.NmgpNumericalError):
        ctx.logpos_svc(bad, g["hyper"])
    with pytest.raises(_lib.NmgpError):
        ctx.set_data(g["x"], np.zeros((8, 9)))              # M > NMGP_MAX_OUTPUTS


def test_primitives_against_reference_golden(ctx):
    g = golden("prims")
    assert np.allclose(ctx.pairwise_distances(g["X1"], g["X2"]), g["pd_12"], rtol=1e-13, atol=1e-13)
    assert np.allclose(ctx.pairwise_distances(g["X1"]), g["pd_11"], rtol=1e-13, atol=1e-13)
    assert np.allclose(ctx.rbf_cov(g["x1"], None, 1.7, 0.4), g["rbf_11"], rtol=1e-13, atol=0)
    assert np.allclose(ctx.rbf_cov(g["x1"], g["x2"], 1.7, 0.4), g["rbf_12"], rtol=1e-13, atol=0)
    assert np.allclose(ctx.rbf_cov(g["X1"], g["X2"], 0.9, 1.3), g["rbf2d_12"], rtol=1e-13, atol=0)
    assert np.allclose(ctx.nonstat_rbf_cov(g["x1"], g["s1"], g["l1"]), g["ns_11"], rtol=1e-13, atol=0)
    assert np.allclose(ctx.nonstat_rbf_cov(g["x1"]), g["ns_11_default"], rtol=1e-13, atol=0)
    assert np.allclose(ctx.nonstat_rbf_cov(g["x1"], g["s1"], g["l1"], g["x2"], g["s2"], g["l2"]), g["ns_12"],
                       rtol=1e-13, atol=0)
    assert np.array_equal(ctx.kron_product(g["B"], g["K"]), g["kron_BK"])
    assert np.array_equal(ctx.kron_product(g["Br"], g["Kr"]), g["kron_rect"])


def test_python_mirror_autograd_matches_reference(ctx):
    """The reference's call pattern (Nonseparable_model.py:163-171): cat -> nlogpos_obj_SVC(verbose) -> backward."""
    import torch
    from nonstationary_multivariate_gaussian_process_amd import Utility
    g = golden("svc_rngfree_N64_M3")
    N, M = g["Y"].shape
    T = M * (M + 1) // 2
    h = hyper_dict(g["hyper"], SVC_KEYS)
    Y, x = torch.from_numpy(g["Y"]), torch.from_numpy(g["x"])
    tilde_l = torch.from_numpy(g["pars"][:N]).clone().requires_grad_(True)
    uL = torch.from_numpy(g["pars"][N:N + N * T]).clone().requires_grad_(True)
    tse = torch.from_numpy(g["pars"][-1:]).clone().requires_grad_(True)
    P = torch.cat([tilde_l, uL, tse.view(1)])
    out = Utility.logpos.nlogpos_obj_SVC(P, Y, x, **h, verbose=True)
    assert len(out) == 5 and all(o.dim() == 0 and o.dtype == torch.float64 for o in out)
    assert relerr([float(o) for o in out], g["out"]) < VAL_TOL
    out[0].backward()
    got = torch.cat([tilde_l.grad, uL.grad, tse.grad]).numpy()
    assert vec_relerr(got, g["grad"]) < GRAD_TOL
    # non-verbose returns the scalar; torch.autograd.grad works on the flat vector too
    p = torch.from_numpy(g["pars"]).clone().requires_grad_(True)
    v = Utility.logpos.nlogpos_obj_SVC(p, Y, x, **h)
    (gp,) = torch.autograd.grad(v, p)
    assert relerr(float(v), g["out"][0]) < VAL_TOL and vec_relerr(gp.numpy(), g["grad"]) < GRAD_TOL
    assert isinstance(float(-v), float)                       # target_value_hist[i] = -NegLog (Nonseparable_model.py:183)
    # a read-only evaluation of a tensor that requires grad (Stationary_model.py:112,168 style): under torch.no_grad() the mirror asks
    # the library for the value alone (no gradient is computed or kept) and returns the same numbers
    with torch.no_grad():
        ro = Utility.logpos.nlogpos_obj_SVC(p, Y, x, **h, verbose=True)
    assert not ro[0].requires_grad and relerr([float(o) for o in ro], g["out"]) < VAL_TOL
    K = Utility.kernels.Nonstationary_RBF_cov(x.view(-1, 1), ell1=torch.exp(tilde_l.detach()))
    assert np.allclose(K.numpy(), g["Kx"], rtol=1e-13, atol=0)


def test_headline_size_properties(ctx):
    """N=2048, D=3 (MN=6144): the committed reference vector plus size-independent identities."""
    g = golden("svc_sim_N2048_M3_base")
    ctx.set_data(g["x"], g["Y"])
    out, grad = ctx.logpos_svc(g["pars"], g["hyper"], prior=True, want_grad=True)
    assert relerr(out[0], g["out"][0]) < VAL_TOL and relerr(out[1], g["out"][1]) < LIK_TOL
    assert vec_relerr(grad, g["grad"]) < GRAD_TOL
    # additivity of the verbose components and the Prior=False identity at full size
    tse = g["pars"][-1]
    assert relerr(out[0], -(out[1] + out[2] + out[3] + out[4] + tse)) < 1e-14
    lik, glik = ctx.logpos_svc(g["pars"], g["hyper"], prior=False, want_grad=True)
    assert lik[0] == -lik[1] == -out[1]
    # directional derivative of the likelihood by central differences along a smooth direction
    rng = np.random.default_rng(0)
    k = np.arange(g["pars"].shape[0])
    v = np.sin(0.003 * k + 0.2) * 1e-2
    eps = 1e-3
    fp, _ = ctx.logpos_svc(g["pars"] + eps * v, g["hyper"], prior=False)
    fm, _ = ctx.logpos_svc(g["pars"] - eps * v, g["hyper"], prior=False)
    fd = (fp[0] - fm[0]) / (2 * eps)
    assert abs(fd - glik @ v) / abs(fd) < 1e-5


# ---------------------------------------------------------------------------------------------------
# separable / stationary objectives, Kronecker primitives, prediction
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("sep_"))
def test_sep_against_reference_golden(ctx, name):
    g = golden(name)
    ctx.set_data(g["x"], g["Y"])
    out, grad = ctx.logpos_sep(g["pars"], g["hyper"], prior=bool(g["prior"]), want_grad=True)
    N = g["Y"].shape[0]
    record_parity(name, neglog=(relerr(out[0], g["out"][0]), VAL_TOL), loglik=(relerr(out[1], g["out"][1]), 1e-8),
                  prior_components_on_the_logdet_scale=(prior_component_err_on_the_logdet_scale(out[2:4], g["out"][2:4], N), VAL_TOL), grad=(vec_relerr(grad, g["grad"]), GRAD_TOL))
    assert relerr(out[0], g["out"][0]) < VAL_TOL, (out, g["out"])
    assert relerr(out[1], g["out"][1]) < 1e-8          # eigen-trick likelihood
    assert prior_component_err_on_the_logdet_scale(out[2:4], g["out"][2:4], N) < VAL_TOL and relerr(out[4:], g["out"][4:]) < VAL_TOL
    assert relerr(out[4], g["out"][4]) < 1e-13         # Normal(0, c) incl. the float32 log(c) quirk
    # the reference backpropagates through eigh, ours is the analytic adjoint; achieved: <= 6e-7 up to N = 512, 4.9e-6 at
    # N = 4096 (the GP-prior part, conditioning-bound: DESIGN.md section 5) -- the same 1e-5 bar as the nonseparable gradient
    assert vec_relerr(grad, g["grad"]) < GRAD_TOL
    out2, _ = ctx.logpos_sep(g["pars"], g["hyper"], prior=bool(g["prior"]), want_grad=False)
    assert relerr(out2, out) < 1e-13


@pytest.mark.parametrize("N,M,seed", [(1, 1, 0), (3, 2, 1), (65, 3, 2), (130, 5, 3), (257, 8, 4)])
def test_sep_against_oracle_ragged_sizes(ctx, N, M, seed):
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    d = sim.simulate_separable(N, M, seed)
    pars = sim.perturb(d["pars_true"], 0.05, 0.2 * seed)
    hv = [sim.HYPER_SEP[k] for k in SEP_KEYS]
    ctx.set_data(d["x"], d["Y"])
    out, grad = ctx.logpos_sep(pars, hv, prior=True, want_grad=True)
    ref, gref = O.nlogpos_obj(pars, d["Y"], d["x"], **sim.HYPER_SEP, verbose=True, grad=True)
    assert relerr(out[0], ref[0]) < VAL_TOL, (out, ref)
    assert relerr(out[1], ref[1]) < 1e-8
    assert vec_relerr(grad, gref) < GRAD_TOL
    lik, glik = ctx.logpos_sep(pars, hv, prior=False, want_grad=True)
    assert lik[0] == -lik[1]


@pytest.mark.parametrize("name", golden_names("sta_"))
def test_sta_against_reference_golden(ctx, name):
    g = golden(name)
    ctx.set_data(g["x"], g["Y"])
    out, grad = ctx.logpos_sta(g["pars"], g["hyper"], prior=True, want_grad=True)
    record_parity(name, neglog=(relerr(out[0], g["out"][0]), 1e-8), components=(relerr(out, g["out"]), 1e-8),
                  grad=(vec_relerr(grad, g["grad"]), 1e-8))
    assert relerr(out, g["out"]) < 1e-8, (out, g["out"])
    assert vec_relerr(grad, g["grad"]) < 1e-8          # achieved <= 1e-10 (no GP prior in this model)
    from oracle import nmgp_oracle as O
    ref, gref = O.nlogpos_obj_S(g["pars"], g["Y"], g["x"], **hyper_dict(g["hyper"], STA_KEYS), verbose=True, grad=True)
    assert vec_relerr(grad, gref) < 1e-7


def test_kron_primitives_against_reference_golden(ctx):
    g = golden("prims")
    assert np.allclose(ctx.kron_mv(g["B"], g["K"], g["yk"]), g["kron_mv_sq"], rtol=1e-12, atol=1e-12)
    assert np.allclose(ctx.kron_mv(g["Br"], g["Kr"], g["yr"]), g["kron_mv_rect"], rtol=1e-12, atol=1e-12)
    z = np.zeros(18)
    s2 = float(g["sig2"])
    assert relerr(ctx.mvn_logpdf_kron(g["yk"], z, g["B"], g["K"], s2), g["logpdf0"]) < 1e-10
    assert relerr(ctx.mvn_logpdf_kron(g["yk"], None, g["B"], g["K"], s2, dense=True), g["logpdf2"]) < 1e-10
    inv, ld = ctx.kron_inv_logdet(s2, g["B"], g["K"])
    assert np.allclose(inv, g["kron_inv"], rtol=1e-9, atol=1e-11) and abs(ld - float(g["kron_logdet"])) < 1e-10
    # the reference's smoke identity (distributions.py:162-169): logpdf0 == logpdf(y, 0, kron_logdet, kron_inv)
    assert relerr(ctx.mvn_logpdf(g["yk"], z, ld, inv), g["logpdf"]) < 1e-10
    # kron_mv == dense kron @ y on a bigger rectangular case (kronecker_operation.py:111-116)
    rng = np.random.default_rng(5)
    B, K, y = rng.standard_normal((5, 7)), rng.standard_normal((300, 513)), rng.standard_normal(7 * 513)
    assert np.allclose(ctx.kron_mv(B, K, y), ctx.kron_product(B, K) @ y, rtol=1e-11, atol=1e-9)
    B12 = rng.standard_normal((3, 12))
    y12 = rng.standard_normal(12 * 513)
    assert np.allclose(ctx.kron_mv(B12, K, y12), np.kron(B12, K) @ y12, rtol=1e-11, atol=1e-9)


def test_python_mirror_sep_sta_and_primitives(ctx):
    import torch
    from nonstationary_multivariate_gaussian_process_amd import Utility as U
    g = golden("sep_rngfree_N64_M3")
    h = hyper_dict(g["hyper"], SEP_KEYS)
    p = torch.from_numpy(g["pars"]).clone().requires_grad_(True)
    out = U.logpos.nlogpos_obj(p, torch.from_numpy(g["Y"]), torch.from_numpy(g["x"]), **h, verbose=True)
    assert len(out) == 6 and relerr([float(o.detach()) for o in out], g["out"]) < VAL_TOL
    out[0].backward()
    assert vec_relerr(p.grad.numpy(), g["grad"]) < GRAD_TOL
    g = golden("sta_rngfree_N64_M3")
    h = hyper_dict(g["hyper"], STA_KEYS)
    p = torch.from_numpy(g["pars"]).clone().requires_grad_(True)
    out = U.logpos.nlogpos_obj_S(p, torch.from_numpy(g["Y"]), torch.from_numpy(g["x"]), h["mu_tilde_l"],
                                 h["sigma_tilde_l"], a=h["a"], b=h["b"], c=h["c"], verbose=True)
    assert len(out) == 5 and relerr([float(o.detach()) for o in out], g["out"]) < 1e-8
    out[0].backward()
    assert vec_relerr(p.grad.numpy(), g["grad"]) < 1e-8
    gp = golden("prims")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    assert np.allclose(U.kronecker_operation.kron_mv(t(gp["Br"]), t(gp["Kr"]), t(gp["yr"])).numpy(), gp["kron_mv_rect"])
    assert np.array_equal(U.kronecker_operation.kronecker_product_diag(t(gp["s1"]), t(gp["l2"])).numpy(), gp["kron_diag"])
    s2 = torch.tensor(float(gp["sig2"]), dtype=torch.float64)
    l0 = U.distributions.multivariate_normal_logpdf0(t(gp["yk"]), torch.zeros(18, dtype=torch.float64), t(gp["B"]),
                                                     t(gp["K"]), s2)
    inv = U.kronecker_operation.kron_inv(s2, t(gp["B"]), t(gp["K"]))
    ld = U.kronecker_operation.kron_logdet(s2, t(gp["B"]), t(gp["K"]))
    l1 = U.distributions.multivariate_normal_logpdf(t(gp["yk"]), torch.zeros(18, dtype=torch.float64), ld, inv)
    assert relerr(float(l0), gp["logpdf0"]) < 1e-10 and relerr(float(l1), float(l0)) < 1e-10
    assert relerr(float(U.distributions.inverse_gamma_logpdf(torch.tensor(0.3, dtype=torch.float64), 2.0, 0.7)),
                  gp["invgamma"]) < 1e-14


def test_separable_chains_batched_equal_single_chain_evaluations(ctx):
    """nmgp_sep_batch_eval: B chains of the separable model as ONE batch of B*M blocks (factorisation, triangular products, inverse
    SYRK batched; the small per-chain pieces queued chain by chain).  Chain 0 carries a reference golden (value + autograd
    gradient), every chain must equal its single-chain evaluation -- on the fused-step schedule (B*M*N small) and on the throughput
    schedule (16 chains x 5 blocks of N = 1024: recursive panels, leaf launches, wide outer panels)."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    for name, B in (("sep_rngfree_N64_M3", 4), ("sep_sim_N512_M5", 3)):
        g = golden(name)
        ctx.set_data(g["x"], g["Y"])
        pars = np.stack([sim.perturb(g["pars"], 0.01 * k, 0.4 * k) for k in range(B)])
        pars[0] = g["pars"]
        out, grad, st = ctx.sep_batch_eval(pars, g["hyper"], bool(g["prior"]), True)
        outv, _, stv = ctx.sep_batch_eval(pars, g["hyper"], bool(g["prior"]), False)
        assert np.all(st == 0) and np.all(stv == 0)
        record_parity("sepbatch_%s_chain0_vs_golden" % name, neglog=(relerr(out[0][0], g["out"][0]), VAL_TOL),
                      grad=(vec_relerr(grad[0], g["grad"]), GRAD_TOL))
        assert relerr(out[0][0], g["out"][0]) < VAL_TOL and vec_relerr(grad[0], g["grad"]) < GRAD_TOL
        for k in range(B):
            so, sg = ctx.logpos_sep(pars[k], g["hyper"], bool(g["prior"]), True)
            assert relerr(out[k][1], so[1]) < 1e-10 and relerr(out[k], so) < 1e-9, (name, k, out[k], so)
            assert relerr(outv[k], so) < 1e-9
            assert vec_relerr(grad[k], sg) < 1e-8, (name, k, vec_relerr(grad[k], sg))
    # throughput schedule
    N, M, B = 1024, 5, 16
    d = sim.simulate_separable(N, M, seed=21)
    hv = [sim.HYPER_SEP[k] for k in SEP_KEYS]
    pars = np.stack([sim.perturb(d["pars_true"], 0.03, 0.3 + 0.2 * k) for k in range(B)])
    ctx.set_data(d["x"], d["Y"])
    out, grad, st = ctx.sep_batch_eval(pars, hv, True, True)
    assert np.all(st == 0) and np.all(np.isfinite(out)) and np.all(np.isfinite(grad))
    for k in (0, 7, 15):
        so, sg = ctx.logpos_sep(pars[k], hv, True, True)
        record_parity("sepbatch16_N1024_M5_chain%d_vs_single" % k, loglik=(relerr(out[k][1], so[1]), 1e-10), grad=(vec_relerr(grad[k], sg), 1e-8))
        assert relerr(out[k][1], so[1]) < 1e-10 and relerr(out[k], so) < 1e-9 and vec_relerr(grad[k], sg) < 1e-8
    # config 5's real shape, 4 chains = 20 blocks of n = 4096 (2048-wide outer panels): chain 0 is the reference golden
    g = golden("sep_sim_N4096_M5")
    ctx.set_data(g["x"], g["Y"])
    pars = np.stack([sim.perturb(g["pars"], 0.01 * k, 0.4 * k) for k in range(4)])
    pars[0] = g["pars"]
    out, grad, st = ctx.sep_batch_eval(pars, g["hyper"], bool(g["prior"]), True)
    assert np.all(st == 0)
    record_parity("sepbatch4_N4096_M5_chain0_vs_golden", neglog=(relerr(out[0][0], g["out"][0]), VAL_TOL), loglik=(relerr(out[0][1], g["out"][1]), 1e-8),
                  grad=(vec_relerr(grad[0], g["grad"]), GRAD_TOL))
    assert relerr(out[0][0], g["out"][0]) < VAL_TOL and relerr(out[0][1], g["out"][1]) < 1e-8 and vec_relerr(grad[0], g["grad"]) < GRAD_TOL
    so, sg = ctx.logpos_sep(pars[3], g["hyper"], bool(g["prior"]), True)
    assert relerr(out[3][1], so[1]) < 1e-10 and relerr(out[3], so) < 1e-9 and vec_relerr(grad[3], sg) < 1e-8
    # a chain whose covariance is numerically singular (the case of test_separable_and_stationary_objectives_recover_from_a_singular_
    # covariance: a zero row of B and sigma2 = 0) goes through the single-chain entry's jitter retries; the others are unaffected
    N, M = 96, 3
    d = sim.simulate_separable(N, M, 3)
    good = d["pars_true"].copy()
    sing = good.copy()
    T = M * (M + 1) // 2
    uL = sing[2 * N:2 * N + T].copy()
    uL[1], uL[2], uL[4] = 0.0, -800.0, 0.0
    sing[2 * N:2 * N + T] = uL
    sing[-1] = -800.0
    ctx.set_data(d["x"], d["Y"])
    out, grad, st = ctx.sep_batch_eval(np.stack([good, sing, good]), hv, False, True)
    s_ok, g_ok = ctx.logpos_sep(good, hv, False, True)
    s_bad, g_bad = ctx.logpos_sep(sing, hv, False, True)
    assert list(st) == [0, 1, 0], st
    assert relerr(out[0], s_ok) < 1e-9 and relerr(out[2], s_ok) < 1e-9 and vec_relerr(grad[0], g_ok) < 1e-8
    assert relerr(out[1][:2], s_bad[:2]) < 1e-12 and np.all(np.isfinite(grad[1])) and vec_relerr(grad[1], g_bad) < 1e-10


def test_prediction_against_reference_golden(ctx):
    """North-star tolerance: predictive mean / variance within 1e-5 of the reference."""
    import torch
    from nonstationary_multivariate_gaussian_process_amd import Utility as U
    g = golden("pred_N64_M3")
    N, M = g["Y"].shape
    T = M * (M + 1) // 2
    ctx.set_data(g["x"], g["Y"])
    mean, var, Ls = ctx.predict_svc(g["svc_pars"], g["svc_hyper"], g["xs"])
    ref = g["svc_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    assert np.allclose(Ls, g["svc_Lstar"], rtol=1e-6, atol=1e-9)
    mean, var = ctx.predict_sep(g["sep_pars"], g["sep_hyper"], g["xs"])
    ref = g["sep_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    mean, var = ctx.predict_sta(g["sta_pars"], g["xs"])
    assert np.allclose(mean, g["sta_mean"], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, g["sta_std"] ** 2, rtol=1e-5, atol=1e-9)
    # through the mirror, with the reference's signatures and return shapes
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    h = hyper_dict(g["svc_hyper"], SVC_KEYS)
    p = g["svc_pars"]
    pct, Lv = U.prediction.point_predmap_inhomogeneous(t(p[:N]), t(p[N:N + N * T]), t(p[-1:])[0], t(g["Y"]), t(g["x"]),
                                                       t(g["xs"])[2], h["mu_tilde_l"], h["alpha_tilde_l"],
                                                       h["beta_tilde_l"], h["mu_L"], h["alpha_L"], h["beta_L"])
    assert pct.shape == (3, M) and Lv.shape == (T,)
    assert np.allclose(pct.numpy(), g["svc_pct"][2], rtol=1e-5, atol=1e-7)
    pw, Lg = U.prediction.pointwise_predmap_inhomogeneous(t(p[:N]), t(p[N:N + N * T]), t(p[-1:])[0], t(g["Y"]), t(g["x"]),
                                                          t(g["xs"]), h["mu_tilde_l"], h["alpha_tilde_l"],
                                                          h["beta_tilde_l"], h["mu_L"], h["alpha_L"], h["beta_L"])
    assert pw.shape == (len(g["xs"]), 3, M) and np.allclose(pw.numpy(), g["svc_pct"], rtol=1e-5, atol=1e-7)
    p = g["sta_pars"]
    mu, sd = U.prediction.test_predmap_S(t(p[:1])[0], t(p[1:2])[0], t(p[2:2 + T]), t(p[-1:])[0], t(g["Y"]), t(g["x"]),
                                         t(g["xs"]))
    assert np.allclose(mu.numpy(), g["sta_mean"], rtol=1e-5, atol=1e-7) and np.allclose(sd.numpy(), g["sta_std"], rtol=1e-5)
    pw = U.prediction.pointwise_predmap_S(t(p[:1])[0], t(p[1:2])[0], t(p[2:2 + T]), t(p[-1:])[0], t(g["Y"]), t(g["x"]),
                                          t(g["xs"]))
    assert np.allclose(pw.numpy(), g["sta_pct"], rtol=1e-5, atol=1e-7)


def test_prediction_on_the_reference_grid(ctx):
    """Prediction at the size the reference's scripts run it: the 201-point grid linspace(0, 1, 201) of Nonseparable_model.py:333
    at N = 512, D = 3, against pointwise_predmap_inhomogeneous / pointwise_predmap / pointwise_predmap_S of the reference
    (tests/golden/pred_N512_M3_grid201.npz); mean / variance within 1e-5."""
    import torch
    from nonstationary_multivariate_gaussian_process_amd import Utility as U
    g = golden("pred_N512_M3_grid201")
    N, M = g["Y"].shape
    T = M * (M + 1) // 2
    xs = g["grids"]
    ctx.set_data(g["x"], g["Y"])
    mean, var, Ls = ctx.predict_svc(g["svc_pars"], g["svc_hyper"], xs)
    ref = g["svc_pct"]
    rv = ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2
    record_parity("pred_N512_M3_grid201_svc", mean=(float(np.max(np.abs(mean - ref[:, 1]) / (np.abs(ref[:, 1]) + 1e-2))), 1e-5),
                  var=(float(np.max(np.abs(var - rv) / rv)), 1e-5))
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, rv, rtol=1e-5, atol=1e-9)
    # L* is a GP regression through RBF(alpha=10, beta=1) + 1e-6 I at N = 512 (condition number ~1e11): the reference solves it
    # by LU (torch.solve), we by Cholesky -- both backward stable, 2e-7 apart in absolute terms on entries of order one
    assert np.allclose(Ls, g["svc_Lstar"], rtol=1e-6, atol=1e-6)
    ctx.set_data(g["sep_x"], g["sep_Y"])
    mean, var = ctx.predict_sep(g["sep_pars"], g["sep_hyper"], xs)
    ref = g["sep_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    ctx.set_data(g["sta_x"], g["sta_Y"])
    mean, var = ctx.predict_sta(g["sta_pars"], xs)
    assert np.allclose(mean, g["sta_mean"], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, g["sta_std"] ** 2, rtol=1e-5, atol=1e-9)
    # through the mirror: the whole grid in one call, the reference's return shapes
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    h = hyper_dict(g["svc_hyper"], SVC_KEYS)
    p = g["svc_pars"]
    pw, Lg = U.prediction.pointwise_predmap_inhomogeneous(t(p[:N]), t(p[N:N + N * T]), t(p[-1:])[0], t(g["Y"]), t(g["x"]), t(xs),
                                                          h["mu_tilde_l"], h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_L"],
                                                          h["alpha_L"], h["beta_L"])
    assert pw.shape == (201, 3, M) and Lg.shape == (201, T)
    assert np.allclose(pw.numpy(), g["svc_pct"], rtol=1e-5, atol=1e-7)


def test_prediction_many_grid_points_on_a_fresh_context():
    """A context that has done nothing else predicts on 201 points (the scratch block of nmgp_predict_svc was sized for <= 16 grid
    points and only survived behind larger earlier allocations); small N, against the oracle."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from oracle import nmgp_oracle as O
    N, M = 64, 3
    x, Y = sim.rngfree_inputs(N, M)
    p = sim.rngfree_pars_svc(N, M)
    h = sim.HYPER_SVC
    xs = np.linspace(0.0, 1.0, 201)
    c = _lib.Context(0)
    try:
        c.set_data(x, Y)
        mean, var, Ls = c.predict_svc(p, [h[k] for k in SVC_KEYS], xs)
    finally:
        c.close()
    tl, uL, tse = O.vec2pars_SVC(p, N, M)
    _, Lo, mo, vo = O.predmap_inhomogeneous(tl, uL, tse, Y, x, xs, h["mu_tilde_l"], h["alpha_tilde_l"], h["beta_tilde_l"],
                                            h["mu_L"], h["alpha_L"], h["beta_L"])
    assert np.allclose(mean, mo, rtol=1e-5, atol=1e-7) and np.allclose(var, vo, rtol=1e-5, atol=1e-9)
    assert np.allclose(Ls, Lo, rtol=1e-6, atol=1e-6)


def test_nonseparable_prediction_at_the_headline_size_against_the_oracle(ctx):
    """predict_svc at N = 2048, D = 3 on the reference's 201-point grid (prediction.py:912-1012 as called from
    Nonseparable_model.py:333) -- the size tools/pred_bench.py times: eight grid points against the CPU oracle (one 6144^2
    factorisation on the host), and the whole grid against the same call made in two slices of the grid, so that results do not
    depend on how many cross-covariance rows ride in one factorisation.  Mean / variance within 1e-5."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    N, M = 2048, 3
    d = sim.simulate_nonseparable(N, M, seed=2222)
    p = sim.perturb(d["pars_true"], 0.05, 0.3)
    h = sim.HYPER_SVC
    hv = [h[k] for k in SVC_KEYS]
    xs = np.linspace(0.0, 1.0, 201)
    ctx.set_data(d["x"], d["Y"])
    mean, var, Ls = ctx.predict_svc(p, hv, xs)
    assert mean.shape == (201, M) and var.shape == (201, M) and np.all(var > 0)
    m1, v1, L1 = ctx.predict_svc(p, hv, xs[:100])
    m2, v2, L2 = ctx.predict_svc(p, hv, xs[100:])
    ms, vs, Lss = np.concatenate([m1, m2]), np.concatenate([v1, v2]), np.concatenate([L1, L2])
    slice_mean = float(np.max(np.abs(mean - ms) / (np.abs(ms) + 1e-2)))
    slice_var = float(np.max(np.abs(var - vs) / vs))
    assert np.allclose(mean, ms, rtol=1e-9, atol=1e-11) and np.allclose(var, vs, rtol=1e-9, atol=1e-12) and np.allclose(Ls, Lss, rtol=1e-9, atol=1e-12)
    pick = np.array([0, 7, 50, 99, 100, 137, 188, 200])
    tl, uL, tse = O.vec2pars_SVC(p, N, M)
    _, Lo, mo, vo = O.predmap_inhomogeneous(tl, uL, tse, d["Y"], d["x"], xs[pick], h["mu_tilde_l"], h["alpha_tilde_l"], h["beta_tilde_l"],
                                            h["mu_L"], h["alpha_L"], h["beta_L"])
    record_parity("pred_svc_N2048_M3_grid201_oracle",
                  mean=(float(np.max(np.abs(mean[pick] - mo) / (np.abs(mo) + 1e-2))), 1e-5),
                  var=(float(np.max(np.abs(var[pick] - vo) / vo)), 1e-5), slices_mean=slice_mean, slices_var=slice_var)
    assert np.allclose(mean[pick], mo, rtol=1e-5, atol=1e-7) and np.allclose(var[pick], vo, rtol=1e-5, atol=1e-9)
    assert np.allclose(Ls[pick], Lo, rtol=1e-6, atol=1e-6)


def test_batch_with_two_different_prior_factors_equals_single_chain_evaluations(ctx):
    """Chains of ONE subject whose GP priors on l~ and on uL have different hyper-parameters (the reference's _distributed
    scripts: alpha 5, beta_tilde_l 0.1, beta_L 0.2, Nonseparable_model_distributed.py:47-48) share two prior factors: the batch
    solves them with two strided-batched calls (factor stride 0).  Every chain must reproduce its single-chain evaluation."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    N, M, B = 160, 3, 5
    d = sim.simulate_nonseparable(N, M, seed=11)
    hv = np.array([0.0, 5.0, 0.1, 0.0, 5.0, 0.2, 1.0, 1.0])
    allp = np.stack([sim.perturb(d["pars_true"], 0.05, 0.3 + 0.5 * b) for b in range(B)])
    ctx.set_data(d["x"], d["Y"])
    singles = [ctx.logpos_svc(allp[b], hv, prior=True, want_grad=True) for b in range(B)]
    c = _lib.Context(0)
    try:
        c.set_data(d["x"], d["Y"])
        c.svc_batch_alloc(B)
        c.svc_batch_set_pars(allp)
        c.svc_batch_eval(hv, True, True)
        out, st = c.svc_batch_fetch()
        grads = c.svc_batch_fetch_grad()
    finally:
        c.close()
    assert np.all(st == 0)
    for b in range(B):
        assert relerr(out[b], singles[b][0]) < 1e-11, (b, out[b], singles[b][0])
        assert vec_relerr(grads[b], singles[b][1]) < 1e-10


def test_separable_prediction_at_config5_size_against_the_oracle(ctx):
    """Config 5's shape (separable, N = 4096, D = 5): the predictor through the Cholesky formulation -- five N x N blocks as one batch,
    the cross-covariance vectors riding as extra rows, GP regressions of l~* and sigma~* by substitution (the right-hand side in
    more than 64 KB of LDS) -- against the CPU oracle's eigen formulation (prediction.py:337-408) on a few grid points; and the
    library's own eigen formulation (NMGP_SEP=eig) on the same inputs."""
    import os
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from oracle import nmgp_oracle as O
    N, M = 4096, 5
    d = sim.simulate_separable(N, M, seed=8)
    p = d["pars_true"].copy()
    p[:N] += 0.05 * np.sin(3.0 * d["x"] + 0.4)
    h = sim.HYPER_SEP
    hv = [h[k] for k in SEP_KEYS]
    xs = np.array([0.0, 0.013, 0.2, 0.37, 0.5, 0.613, 0.88, 1.0])
    ctx.set_data(d["x"], d["Y"])
    mean, var = ctx.predict_sep(p, hv, xs)
    tl, ts, uLv, tse = O.vec2pars(p, N, M)
    _, mo, vo = O.predmap_separable(tl, ts, uLv, tse, d["Y"], d["x"], xs, h["mu_tilde_l"], h["alpha_tilde_l"], h["beta_tilde_l"],
                                    h["mu_tilde_sigma"], h["alpha_tilde_sigma"], h["beta_tilde_sigma"])
    record_parity("pred_sep_N4096_M5_oracle", mean=(float(np.max(np.abs(mean - mo) / (1e-7 + 1e-5 * np.abs(mo)))) * 1e-5, 1e-5),
                  var=(float(np.max(np.abs(var - vo) / vo)), 1e-5))
    assert np.allclose(mean, mo, rtol=1e-5, atol=1e-7) and np.allclose(var, vo, rtol=1e-5, atol=1e-9)
    os.environ["NMGP_SEP"] = "eig"
    try:
        c2 = _lib.Context(0)
    finally:
        os.environ.pop("NMGP_SEP", None)
    try:
        c2.set_data(d["x"], d["Y"])
        m2, v2 = c2.predict_sep(p, hv, xs)
    finally:
        c2.close()
    assert np.allclose(mean, m2, rtol=1e-7, atol=1e-9) and np.allclose(var, v2, rtol=1e-6, atol=1e-10)


# ---------------------------------------------------------------------------------------------------
# custom blocked Cholesky (nmgp_chol.hip): FP64-MFMA SYRK, 64-wide panel steps, right-hand side as an extra row
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 17, 63, 64, 65, 127, 130, 257, 513, 600, 1000, 1537, 1600, 2112])
def test_custom_cholesky_against_lapack(ctx, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T / n + 0.5 * np.eye(n)
    # asymmetric scaling so that a row/column mix-up in the MFMA fragment layout cannot cancel out
    dscale = 1.0 + np.arange(n) / n
    A = A * np.outer(dscale, dscale)
    rhs = rng.standard_normal(n)
    Lref = np.linalg.cholesky(A)
    zref = np.linalg.solve(Lref, rhs)
    for algo in (1, 0):
        L, z = ctx.cholesky(A, rhs, algo=algo)
        assert np.allclose(L, Lref, rtol=1e-11, atol=1e-12), (algo, n, np.abs(L - Lref).max())
        assert np.allclose(z, zref, rtol=1e-10, atol=1e-11), (algo, n)
    L2 = ctx.cholesky(A, None, algo=1)
    assert np.array_equal(L2, ctx.cholesky(A, rhs, algo=1)[0])


# (n, bad pivot): inside the first panel; the first block of the second panel (factored inside the look-ahead's near update);
# a later block of the second panel (fused panel step); the ragged last panel
@pytest.mark.parametrize("n,bad", [(200, 150), (1200, 520), (1200, 700), (1200, 1100)])
def test_custom_cholesky_reports_indefinite_matrix(ctx, n, bad):
    from nonstationary_multivariate_gaussian_process_amd import _lib
    rng = np.random.default_rng(n + bad)
    G = rng.standard_normal((n, 8)) / np.sqrt(n)
    A = np.eye(n) + G @ G.T
    A[bad, bad] = -1.0
    with pytest.raises(_lib.NmgpNumericalError) as e:
        ctx.cholesky(A, None, algo=1)
    assert e.value.code == bad + 1      # LAPACK convention: leading minor bad + 1 is not positive definite


def test_rocsolver_and_custom_factorisation_agree_on_the_objective():
    import os
    from nonstationary_multivariate_gaussian_process_amd import _lib
    g = golden("svc_sim_N1024_M3_base")
    res, pred = {}, {}
    for algo in ("custom", "rocsolver"):
        os.environ["NMGP_CHOL"] = algo
        try:
            c = _lib.Context(0)
        finally:
            os.environ.pop("NMGP_CHOL", None)
        c.set_data(g["x"], g["Y"])
        res[algo] = c.logpos_svc(g["pars"], g["hyper"], prior=True, want_grad=True)
        # prediction through the same context: custom = the cross-covariance rows ride the matrix-core factorisation,
        # rocsolver = library dpotrf + one right-sided dtrsm on those rows
        gp = golden("pred_N64_M3")
        c.set_data(gp["x"], gp["Y"])
        pred[algo] = c.predict_svc(gp["svc_pars"], gp["svc_hyper"], gp["xs"])
        c.close()
    ref = gp["svc_pct"]
    for algo in pred:
        mean, var, Ls = pred[algo]
        assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7), algo
        assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9), algo
    assert np.allclose(pred["custom"][0], pred["rocsolver"][0], rtol=1e-9, atol=1e-11)
    # the likelihood agrees to rounding; the GP-prior terms (condition number ~1e11) agree to their conditioning noise,
    # and both stay within the 1e-6 target of the reference (tools/prior_accuracy.py)
    assert relerr(res["custom"][0][1], res["rocsolver"][0][1]) < 1e-11
    assert relerr(res["custom"][0], res["rocsolver"][0]) < VAL_TOL
    assert vec_relerr(res["custom"][1], res["rocsolver"][1]) < GRAD_TOL
    assert relerr(res["custom"][0], g["out"]) < VAL_TOL and relerr(res["rocsolver"][0], g["out"]) < VAL_TOL


def test_batched_chains_match_single_evaluations(ctx):
    """B chains per launch sequence: every row equals the single-chain evaluation; a broken chain is isolated."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    for name, B in (("svc_rngfree_N64_M3", 5), ("svc_sim_N1024_M3_base", 3), ("svc_sim_N77_M2_base", 4)):
        g = golden(name)
        ctx.set_data(g["x"], g["Y"])
        ctx.svc_batch_alloc(B)
        pars = np.stack([sim.perturb(g["pars"], 0.02 * k, 0.3 * k) for k in range(B)])
        pars[0] = g["pars"]
        ctx.svc_batch_set_pars(pars)
        ctx.svc_batch_eval(g["hyper"], True)
        out, status = ctx.svc_batch_fetch()
        assert np.all(status == 0)
        assert relerr(out[0], g["out"]) < VAL_TOL and relerr(out[0][1], g["out"][1]) < LIK_TOL
        ctx.svc_batch_eval(g["hyper"], True, want_grad=True)
        outg, statusg = ctx.svc_batch_fetch()
        grads = ctx.svc_batch_fetch_grad()
        assert np.all(statusg == 0) and relerr(outg, out) < 1e-9
        assert vec_relerr(grads[0], g["grad"]) < GRAD_TOL
        for k in range(B):
            single, gsingle = ctx.logpos_svc(pars[k], g["hyper"], prior=True, want_grad=True)
            assert relerr(out[k][1], single[1]) < 1e-12 and relerr(out[k], single) < 1e-9, (name, k, out[k], single)
            assert vec_relerr(grads[k], gsingle) < 1e-9
    g = golden("svc_rngfree_N64_M3")
    ctx.set_data(g["x"], g["Y"])
    ctx.svc_batch_alloc(3)
    pars = np.stack([g["pars"]] * 3)
    pars[1, 5] = np.nan
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(g["hyper"], True)
    out, status = ctx.svc_batch_fetch()
    assert status[0] == 0 and status[2] == 0 and status[1] != 0
    assert np.all(np.isnan(out[1])) and relerr(out[0], g["out"]) < VAL_TOL and np.array_equal(out[0], out[2])


def test_large_batches_at_full_size_match_the_golden_vector_and_single_evaluations(ctx):
    """The benchmark's configuration in small: 16 chains of the N = 2048, D = 3 subject per launch sequence take the widest
    outer panels (2048) and the recursive panel factorisation; 4 chains take 1024.  Chain 0 carries the golden parameters
    (reference output committed under tests/golden), the others are checked against single-chain evaluations, which
    factor with 512-wide panels and right-looking 64-wide steps."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_sim_N2048_M3_base")
    ctx.set_data(g["x"], g["Y"])
    for B in (16, 4):
        ctx.svc_batch_alloc(B)
        pars = np.stack([sim.perturb(g["pars"], 0.01 * k, 0.2 * k) for k in range(B)])
        pars[0] = g["pars"]
        ctx.svc_batch_set_pars(pars)
        ctx.svc_batch_eval(g["hyper"], True)
        out, status = ctx.svc_batch_fetch()
        assert np.all(status == 0)
        assert relerr(out[0], g["out"]) < VAL_TOL and relerr(out[0][1], g["out"][1]) < LIK_TOL
        for k in (1, B // 2, B - 1):
            single = ctx.logpos_svc(pars[k], g["hyper"], prior=True, want_grad=False)[0]
            assert relerr(out[k][1], single[1]) < 1e-11 and relerr(out[k], single) < 1e-9, (B, k, out[k], single)
    # gradients of the 4-chain batch (L^-T rows ride along, 1024-wide panels) against the golden gradient / a single evaluation
    ctx.svc_batch_eval(g["hyper"], True, want_grad=True)
    outg, statusg = ctx.svc_batch_fetch()
    grads = ctx.svc_batch_fetch_grad()
    assert np.all(statusg == 0) and relerr(outg, out) < 1e-9
    assert vec_relerr(grads[0], g["grad"]) < GRAD_TOL
    gsingle = ctx.logpos_svc(pars[3], g["hyper"], prior=True, want_grad=True)[1]
    assert vec_relerr(grads[3], gsingle) < 1e-8
    ctx.svc_batch_alloc(1)      # release the batch buffers


def test_separable_cholesky_and_eigen_formulations_agree():
    """The default separable/stationary path factors M blocks wB[p] K + sigma2 I with the batched Cholesky; the
    reference's eigen-trick formulation (distributions.py:26-52) stays selectable and must give the same numbers."""
    import os
    from nonstationary_multivariate_gaussian_process_amd import _lib
    res = {}
    for algo in ("chol", "eig"):
        os.environ["NMGP_SEP"] = algo
        try:
            c = _lib.Context(0)
        finally:
            os.environ.pop("NMGP_SEP", None)
        for name in ("sep_sim_N512_M5", "sta_sim_N128_M2"):
            g = golden(name)
            c.set_data(g["x"], g["Y"])
            fn = c.logpos_sep if name.startswith("sep") else c.logpos_sta
            res[(algo, name)] = fn(g["pars"], g["hyper"], True, True)
        c.close()
    for name in ("sep_sim_N512_M5", "sta_sim_N128_M2"):
        a, b = res[("chol", name)], res[("eig", name)]
        assert relerr(a[0][1], b[0][1]) < 1e-9 and relerr(a[0], b[0]) < 1e-9
        assert vec_relerr(a[1], b[1]) < 1e-7


@pytest.mark.parametrize("N,M,B", [(96, 3, 5), (77, 3, 6), (129, 2, 11)])
def test_multi_subject_batch_matches_per_subject_evaluations(ctx, N, M, B):
    """BASELINE config 4 pattern: several subjects (own x, Y, own prior factors) in ONE launch sequence; odd N exercises the
    row-pair tail of the prior solves (k_prior_trsv) and the ragged last block of the factorisation."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    subs = [sim.simulate_nonseparable(N, M, seed=s) for s in range(B)]
    hyper = sim.HYPER_SVC_MPISIM
    hv = [hyper[k] for k in SVC_KEYS]
    pars = np.stack([sim.perturb(d["pars_true"], 0.03, 0.2 * k) for k, d in enumerate(subs)])
    ctx.set_data(subs[0]["x"], subs[0]["Y"])
    ctx.svc_batch_alloc(B)
    ctx.svc_batch_set_subjects(np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs]))
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(hv, True, want_grad=True)
    out, status = ctx.svc_batch_fetch()
    grads = ctx.svc_batch_fetch_grad()
    assert np.all(status == 0)
    for k, d in enumerate(subs):
        ctx.set_data(d["x"], d["Y"])
        single, gsingle = ctx.logpos_svc(pars[k], hv, prior=True, want_grad=True)
        assert relerr(out[k][1], single[1]) < 1e-11 and relerr(out[k], single) < 1e-8, (k, out[k], single)
        assert vec_relerr(grads[k], gsingle) < 1e-8

@pytest.mark.gpu
@pytest.mark.parametrize("cond", [1e4, 1e8, 1e11])
def test_custom_cholesky_backward_error_on_ill_conditioned_matrices(ctx, cond):
    """The matrix-core panel kernels multiply by inverted 16x16 diagonal blocks (k_potf2_64b / k_trsm_64m): the
    factorisation must stay backward stable, i.e. ||L L^T - A|| / ||A|| at rounding level whatever the conditioning."""
    n = 333                                          # 5 blocks of 64 + a ragged one of 13
    rng = np.random.default_rng(int(np.log10(cond)))
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.logspace(0, -np.log10(cond), n)
    A = (Q * w) @ Q.T
    A = 0.5 * (A + A.T)
    L = ctx.cholesky(A, None, algo=1)
    res = np.abs(L @ L.T - A).max() / np.abs(A).max()
    Lref = np.linalg.cholesky(A)
    ref = np.abs(Lref @ Lref.T - A).max() / np.abs(A).max()
    assert res < 50 * np.finfo(float).eps * n ** 0.5, (cond, res, ref)
    assert res < 20 * max(ref, np.finfo(float).eps), (cond, res, ref)





# ---------------------------------------------------------------------------------------------------
# BASELINE configs at their real shapes, against the reference's own numbers
# ---------------------------------------------------------------------------------------------------
def test_config4_eight_subjects_N1024_one_batch_against_reference_golden(ctx):
    """BASELINE config 4's per-GPU shape: 8 subjects (own x, Y, own prior factors), N = 1024, D = 3, mpisim hyper-parameters
    (Nonseparable_model_mpisim.py:305-312) evaluated by ONE multi-subject launch sequence (nmgp_svc_batch_set_subjects), value
    and gradient, against what the reference itself returned for each subject (tests/golden/cfg4_subjects_N1024_M3.npz:
    values + autograd gradients) -- at the parameters bench.py's subjects workload evaluates and at the generating ones."""
    g = golden("cfg4_subjects_N1024_M3")
    B, N = g["xs"].shape
    ctx.set_data(g["xs"][0], g["Ys"][0])
    ctx.svc_batch_alloc(B)
    ctx.svc_batch_set_subjects(g["xs"], g["Ys"])
    for pk, ok, gk in (("pars", "out", "grad"), ("pars_true", "out_true", "grad_true")):
        ctx.svc_batch_set_pars(g[pk])
        ctx.svc_batch_eval(g["hyper"], True, want_grad=False)
        out_v, status_v = ctx.svc_batch_fetch()
        ctx.svc_batch_eval(g["hyper"], True, want_grad=True)
        out, status = ctx.svc_batch_fetch()
        grads = ctx.svc_batch_fetch_grad()
        assert np.all(status == 0) and np.all(status_v == 0)
        for s in range(B):
            ref = g[ok][s]
            errs = dict(neglog=(relerr(out[s][0], ref[0]), VAL_TOL), loglik=(relerr(out[s][1], ref[1]), LIK_TOL),
                        prior_components_on_the_logdet_scale=(prior_component_err_on_the_logdet_scale(out[s][2:4], ref[2:4], N), VAL_TOL),
                        grad=(vec_relerr(grads[s], g[gk][s]), GRAD_TOL),
                        value_only_vs_grad_path=(relerr(out_v[s][0], out[s][0]), 1e-9))
            record_parity("cfg4_subject%d_%s" % (s, pk), **errs)
            for k, (e, tol) in errs.items():
                assert e < tol, (s, pk, k, e, tol, out[s], ref)
            assert relerr(out[s][4], ref[4]) < 1e-12
    ctx.svc_batch_alloc(1)


def test_config5_separable_N4096_D5_both_formulations_against_reference_golden():
    """BASELINE config 5 at its real shape (logpos.py:216-296, N = 4096, D = 5: the size that selects 1024-wide outer panels
    in the batched block factorisation) through the default Cholesky-block formulation and through the reference's
    eigendecomposition formulation (NMGP_SEP=eig)."""
    import os
    from nonstationary_multivariate_gaussian_process_amd import _lib
    g = golden("sep_sim_N4096_M5")
    N = g["Y"].shape[0]
    res = {}
    for algo in ("chol", "eig"):
        os.environ["NMGP_SEP"] = algo
        try:
            c = _lib.Context(0)
        finally:
            os.environ.pop("NMGP_SEP", None)
        c.set_data(g["x"], g["Y"])
        out, grad = c.logpos_sep(g["pars"], g["hyper"], True, True)
        out_v, _ = c.logpos_sep(g["pars"], g["hyper"], True, False)
        c.close()
        res[algo] = (out, grad)
        errs = dict(neglog=(relerr(out[0], g["out"][0]), VAL_TOL), loglik=(relerr(out[1], g["out"][1]), 1e-8),
                    prior_components_on_the_logdet_scale=(prior_component_err_on_the_logdet_scale(out[2:4], g["out"][2:4], N), VAL_TOL),
                    grad=(vec_relerr(grad, g["grad"]), GRAD_TOL), value_only_vs_grad_path=(relerr(out_v, out), 1e-11))
        record_parity("sep_sim_N4096_M5_" + algo, **errs)
        for k, (e, tol) in errs.items():
            assert e < tol, (algo, k, e, tol, out, g["out"])
    assert relerr(res["chol"][0][1], res["eig"][0][1]) < 1e-9
    assert vec_relerr(res["chol"][1], res["eig"][1]) < 1e-6


@pytest.mark.parametrize("N,M,B", [(192, 3, 130), (213, 3, 120), (320, 2, 118), (256, 3, 100)])
def test_throughput_schedule_leaves_match_single_evaluations(ctx, N, M, B):
    """Batches with more than 73,728 rows in all take the throughput schedule, whose 128-column pieces are two leaf launches
    (k_panel_step<1>, <2>: the K = 64 update folded into the second solve).  Value and gradient (the rows of L^-T enter the
    leaves panel by panel) of such batches against single-chain evaluations, which take the fused 64-column steps: n = 576
    (512 + a 64-column remainder), 639 (a last panel that is no multiple of 64: recursive fallback), 640 (512 + one leaf), 768."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    assert B * N * M > 73728
    d = sim.simulate_nonseparable(N, M, seed=100 + N)
    hv = [sim.HYPER_SVC[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")]
    ctx.set_data(d["x"], d["Y"])
    ctx.svc_batch_alloc(B)
    pars = np.stack([sim.perturb(d["pars_true"], 0.05, 0.1 + 0.003 * k) for k in range(B)])
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(hv, True)
    out, status = ctx.svc_batch_fetch()
    assert np.all(status == 0) and np.all(np.isfinite(out))
    ctx.svc_batch_eval(hv, True, want_grad=True)
    outg, statusg = ctx.svc_batch_fetch()
    grads = ctx.svc_batch_fetch_grad()
    assert np.all(statusg == 0) and relerr(outg, out) < 1e-9
    for k in (0, B // 3, B - 1):
        single, gsingle = ctx.logpos_svc(pars[k], hv, prior=True, want_grad=True)
        assert relerr(out[k][1], single[1]) < 1e-11 and relerr(out[k], single) < 1e-9, (k, out[k], single)
        assert vec_relerr(grads[k], gsingle) < 1e-8, k
    ctx.svc_batch_alloc(1)


def test_headline_batch_of_128_chains_at_N2048(ctx):
    """The configuration bench.py times: 128 chains of the N = 2048, D = 3 subject in one launch sequence (2048-wide outer
    panels, recursive panel factorisation).  Chain 0 carries the golden parameters (reference value committed), three other
    chains are compared with single-chain evaluations, every status is 0 and every row is finite."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_sim_N2048_M3_base")
    ctx.set_data(g["x"], g["Y"])
    B = 128
    ctx.svc_batch_alloc(B)
    pars = np.stack([sim.perturb(g["pars"], 0.002 * k, 0.37 * k) for k in range(B)])
    pars[0] = g["pars"]
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(g["hyper"], True)
    out, status = ctx.svc_batch_fetch()
    assert np.all(status == 0) and np.all(np.isfinite(out))
    record_parity("batch128_chain0_vs_golden", neglog=(relerr(out[0][0], g["out"][0]), VAL_TOL),
                  loglik=(relerr(out[0][1], g["out"][1]), LIK_TOL), priors=(relerr(out[0][2:], g["out"][2:]), VAL_TOL))
    assert relerr(out[0], g["out"]) < VAL_TOL and relerr(out[0][1], g["out"][1]) < LIK_TOL
    for k in (1, 63, 127):
        single = ctx.logpos_svc(pars[k], g["hyper"], prior=True, want_grad=False)[0]
        record_parity("batch128_chain%d_vs_single" % k, loglik=(relerr(out[k][1], single[1]), 1e-11),
                      neglog=(relerr(out[k][0], single[0]), 1e-9))
        assert relerr(out[k][1], single[1]) < 1e-11 and relerr(out[k], single) < 1e-9, (k, out[k], single)
    ctx.svc_batch_alloc(1)


def test_sixteen_chain_gradient_batch_on_the_wide_panel_schedule(ctx):
    """A value+gradient batch on the schedule the headline's gradient step takes (16 chains of n = 6144: 2048-wide outer panels,
    leaf launches, L^-T rows entering panel by panel from the seeded band -- also under NMGP_POISON, which the poison run of this
    file provides): chain 0 carries the golden parameters (reference value and autograd gradient committed), two other chains are
    compared with single-chain evaluations (512-wide panels, fused steps)."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_sim_N2048_M3_base")
    ctx.set_data(g["x"], g["Y"])
    B = 16
    ctx.svc_batch_alloc(B)
    pars = np.stack([sim.perturb(g["pars"], 0.002 * k, 0.37 * k) for k in range(B)])
    pars[0] = g["pars"]
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(g["hyper"], True, True)
    out, status = ctx.svc_batch_fetch()
    grads = ctx.svc_batch_fetch_grad()
    assert np.all(status == 0) and np.all(np.isfinite(out)) and np.all(np.isfinite(grads))
    record_parity("gradbatch16_chain0_vs_golden", neglog=(relerr(out[0][0], g["out"][0]), VAL_TOL),
                  grad=(vec_relerr(grads[0], g["grad"]), GRAD_TOL))
    assert relerr(out[0], g["out"]) < VAL_TOL and vec_relerr(grads[0], g["grad"]) < GRAD_TOL
    for k in (5, 15):
        so, sg = ctx.logpos_svc(pars[k], g["hyper"], prior=True, want_grad=True)
        record_parity("gradbatch16_chain%d_vs_single" % k, loglik=(relerr(out[k][1], so[1]), 1e-11), grad=(vec_relerr(grads[k], sg), 1e-8))
        assert relerr(out[k][1], so[1]) < 1e-11 and vec_relerr(grads[k], sg) < 1e-8, (k, vec_relerr(grads[k], sg))
    ctx.svc_batch_alloc(1)


def test_headline_gradient_batch_of_128_chains_at_N2048(ctx):
    """The shape bench.py's `grad` object and the 128-chain HMC measurement time (Nonseparable_model.py:169-171: value, then
    backward): 128 chains x n = 6144 value+gradient in one launch sequence -- 128 L^-T row blocks riding through the 2048-wide
    panels, the inverse SYRK of 128 matrices, the adjoint pass with blockIdx.z = 128.  Chain 0 carries the golden parameters
    (reference value + autograd gradient committed), three other chains are compared with single-chain evaluations, every status
    is 0 and every number finite."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_sim_N2048_M3_base")
    ctx.set_data(g["x"], g["Y"])
    B = 128
    ctx.svc_batch_alloc(B)
    pars = np.stack([sim.perturb(g["pars"], 0.002 * k, 0.37 * k) for k in range(B)])
    pars[0] = g["pars"]
    ctx.svc_batch_set_pars(pars)
    ctx.svc_batch_eval(g["hyper"], True, True)
    out, status = ctx.svc_batch_fetch()
    grads = ctx.svc_batch_fetch_grad()
    assert np.all(status == 0) and np.all(np.isfinite(out)) and np.all(np.isfinite(grads))
    record_parity("gradbatch128_chain0_vs_golden", neglog=(relerr(out[0][0], g["out"][0]), VAL_TOL),
                  grad=(vec_relerr(grads[0], g["grad"]), GRAD_TOL))
    assert relerr(out[0], g["out"]) < VAL_TOL and vec_relerr(grads[0], g["grad"]) < GRAD_TOL
    for k in (1, 63, 127):
        so, sg = ctx.logpos_svc(pars[k], g["hyper"], prior=True, want_grad=True)
        record_parity("gradbatch128_chain%d_vs_single" % k, loglik=(relerr(out[k][1], so[1]), 1e-11), grad=(vec_relerr(grads[k], sg), 1e-8))
        assert relerr(out[k][1], so[1]) < 1e-11 and relerr(out[k], so) < 1e-9 and vec_relerr(grads[k], sg) < 1e-8, (k, vec_relerr(grads[k], sg))
    ctx.svc_batch_alloc(1)


def test_logpdf1_seeded_jitter_against_reference_golden(ctx):
    """a10 (distributions.py:55-96): the jitter comes from torch.rand, B first, then K; torch.manual_seed makes the
    reference's value reproducible and the mirror draws from the same stream in the same order."""
    import torch
    from nonstationary_multivariate_gaussian_process_amd import Utility as U
    g = golden("prims_logpdf1")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for k in range(int(g["ncases"])):
        y = g["y%d" % k]
        s2 = torch.tensor(float(g["sig2_%d" % k]), dtype=torch.float64)
        torch.manual_seed(int(g["seed%d" % k]))
        v = float(U.distributions.multivariate_normal_logpdf1(t(y), torch.zeros(len(y), dtype=torch.float64), t(g["B%d" % k]),
                                                              t(g["K%d" % k]), s2))
        v0 = float(U.distributions.multivariate_normal_logpdf0(t(y), torch.zeros(len(y), dtype=torch.float64),
                                                               t(g["B%d" % k]), t(g["K%d" % k]), s2))
        record_parity("logpdf1_case%d" % k, logpdf1=(relerr(v, g["logpdf1_%d" % k]), 1e-9),
                      logpdf0=(relerr(v0, g["logpdf0_%d" % k]), 1e-9))
        assert relerr(v, g["logpdf1_%d" % k]) < 1e-9, (k, v, g["logpdf1_%d" % k])
        assert relerr(v0, g["logpdf0_%d" % k]) < 1e-9
        # and through the C ABI with the stored draws
        Bj = g["B%d" % k] + np.diag(g["jitterB%d" % k] * 1e-6)
        Kj = g["K%d" % k] + np.diag(g["jitterK%d" % k] * 1e-6)
        assert relerr(ctx.mvn_logpdf_kron(y, None, Bj, Kj, float(s2)), g["logpdf1_%d" % k]) < 1e-9


def test_separable_and_stationary_objectives_recover_from_a_singular_covariance(ctx):
    """The reference retries a NaN likelihood with `precision`-sized jitter on the diagonals of B and K (logpos.py:267-268,
    436-437, distributions.py:55-96) so that a MAP loop never sees NaN; the library does the same deterministically
    (attempt * 1e-6).  B with a vanishing eigenvalue and sigma2_err = exp(-800) = 0 make the first attempt fail; the
    second one must match the oracle's likelihood of the jittered covariance, and the gradient must be finite."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    N, M = 96, 3
    d = sim.simulate_separable(N, M, 3)
    pars = d["pars_true"].copy()
    T = M * (M + 1) // 2
    uL = pars[2 * N:2 * N + T].copy()
    uL[1], uL[2], uL[4] = 0.0, -800.0, 0.0     # L_10 = 0, L_11 = exp(-800) = 0, L_21 = 0: row/column 1 of B = L L^T is exactly zero
    pars[2 * N:2 * N + T] = uL
    pars[-1] = -800.0
    hv = [sim.HYPER_SEP[k] for k in SEP_KEYS]
    ctx.set_data(d["x"], d["Y"])
    # Prior=False: with sigma2_err = 0 the inverse-gamma prior term is inf - inf in the reference too; the likelihood is the point
    out, grad = ctx.logpos_sep(pars, hv, prior=False, want_grad=True)
    assert np.all(np.isfinite(out[:2])) and out[0] == -out[1] and np.all(np.isfinite(grad))
    assert ctx.last_sep_attempts() == 1          # the caller can tell: value and gradient are those of the jittered covariance
    # the likelihood of attempt 1: B + 1e-6 I, K + 1e-6 I
    Lm = O.vec2lowtriangle(O.uLvec2Lvec(uL, M), M)
    Bf = Lm @ Lm.T + 1e-6 * np.eye(M)
    Kx = O.Nonstationary_RBF_cov(d["x"].reshape(-1, 1), np.exp(pars[N:2 * N]), np.exp(pars[:N])) + 1e-6 * np.eye(N)
    y = d["Y"].T.reshape(-1)
    ref = O.multivariate_normal_logpdf0(y, np.zeros_like(y), Bf, Kx, 0.0)
    record_parity("sep_singular_retry", loglik=(relerr(out[1], ref), 1e-6))
    assert relerr(out[1], ref) < 1e-6, (out[1], ref)
    # a well-posed evaluation is untouched by the retry logic (attempt 0 succeeds): same numbers as the golden vector
    g = golden("sep_rngfree_N64_M3")
    ctx.set_data(g["x"], g["Y"])
    o2, _ = ctx.logpos_sep(g["pars"], g["hyper"], True, False)
    assert relerr(o2[0], g["out"][0]) < VAL_TOL
    assert ctx.last_sep_attempts() == 0


def test_repeated_evaluations_are_bit_identical_on_the_multi_stream_schedules(ctx):
    """The latency-regime schedules run three streams at once (panel steps, far trailing updates on the CU-masked stream, prior
    solves) and hand blocks from one launch to the next through events; every reduction has a fixed order.  A missing
    dependency would show as run-to-run differences long before it shows as a parity failure: the same inputs, evaluated five
    times each (value + gradient), must give the same bits -- one chain at N = 2048 (12 panels: look-ahead, small-tile near
    update with the fused first block), 8 subjects x N = 1024 in one batch, separable N = 4096 x D = 5."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    hv = [sim.HYPER_SVC[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")]
    d = sim.simulate_nonseparable(2048, 3, seed=2222)
    pars = sim.perturb(d["pars_true"], 0.05, 0.3)
    ctx.set_data(d["x"], d["Y"])
    runs = [ctx.logpos_svc(pars, hv, prior=True, want_grad=True) for _ in range(5)]
    for out, grad in runs[1:]:
        assert np.array_equal(out, runs[0][0]) and np.array_equal(grad, runs[0][1])
    vals = [ctx.logpos_svc(pars, hv, prior=True, want_grad=False)[0] for _ in range(5)]
    assert all(np.array_equal(v, vals[0]) for v in vals[1:])
    # 8 subjects in one batch
    subs = [sim.simulate_nonseparable(1024, 3, seed=s) for s in range(8)]
    ctx.set_data(subs[0]["x"], subs[0]["Y"])
    ctx.svc_batch_alloc(8)
    ctx.svc_batch_set_subjects(np.stack([s["x"] for s in subs]), np.stack([s["Y"] for s in subs]))
    ctx.svc_batch_set_pars(np.stack([sim.perturb(s["pars_true"], 0.05, 0.3) for s in subs]))
    hm = [sim.HYPER_SVC_MPISIM[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")]
    res = []
    for _ in range(5):
        ctx.svc_batch_eval(hm, True, want_grad=True)
        o, st = ctx.svc_batch_fetch()
        res.append((o.copy(), ctx.svc_batch_fetch_grad().copy()))
        assert not st.any()
    for o, g in res[1:]:
        assert np.array_equal(o, res[0][0]) and np.array_equal(g, res[0][1])
    ctx.svc_batch_alloc(1)
    # separable, config 5's shape
    ds = sim.simulate_separable(4096, 5, 8)
    ps = sim.perturb(ds["pars_true"], 0.05, 0.4)
    hs = [sim.HYPER_SEP[k] for k in ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma",
                                     "beta_tilde_sigma", "a", "b", "c")]
    ctx.set_data(ds["x"], ds["Y"])
    sr = [ctx.logpos_sep(ps, hs, True, True) for _ in range(5)]
    for o, g in sr[1:]:
        assert np.array_equal(o, sr[0][0]) and np.array_equal(g, sr[0][1])


def test_twice_the_headline_size_against_the_oracle(ctx):
    """N = 4096, D = 3 (MN = 12288, 24 panels): twice BASELINE's largest single-subject matrix.  The k-loop of the inverse SYRK is
    then longer than the 32-bit panel offsets of the mask-free tile path allow (it falls back to the generic path), the look-ahead
    schedule runs 24 panels, the triangular alpha product 48 x 48 blocks: value, components and full gradient against the CPU
    oracle, plus a central-difference check of the gradient along a smooth direction."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O
    N, M = 4096, 3
    d = sim.simulate_nonseparable(N, M, 11)
    pars = sim.perturb(d["pars_true"], 0.05, 0.2)
    hv = [sim.HYPER_SVC[k] for k in SVC_KEYS]
    ctx.set_data(d["x"], d["Y"])
    out, grad = ctx.logpos_svc(pars, hv, prior=True, want_grad=True)
    val, _ = ctx.logpos_svc(pars, hv, prior=True, want_grad=False)
    ref, gref = O.nlogpos_obj_SVC(pars, d["Y"], d["x"], **sim.HYPER_SVC, verbose=True, grad=True)
    errs = dict(neglog=(relerr(out[0], ref[0]), VAL_TOL), loglik=(relerr(out[1], ref[1]), LIK_TOL),
                prior_components_on_the_logdet_scale=(prior_component_err_on_the_logdet_scale(out[2:4], np.array(ref[2:4]), N), VAL_TOL), grad=(vec_relerr(grad, gref), GRAD_TOL),
                value_only_vs_grad_path=(relerr(val[0], out[0]), 1e-9))
    record_parity("svc_sim_N4096_M3_oracle", **errs)
    for k, (e, tol) in errs.items():
        assert e < tol, (k, e, tol, out, ref)
    k = np.arange(pars.shape[0])
    v = np.sin(0.002 * k + 0.1) * 1e-2
    eps = 1e-3
    fp, _ = ctx.logpos_svc(pars + eps * v, hv, prior=True)
    fm, _ = ctx.logpos_svc(pars - eps * v, hv, prior=True)
    fd = (fp[0] - fm[0]) / (2 * eps)
    assert abs(fd - grad @ v) / abs(fd) < 1e-5
