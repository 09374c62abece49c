"""The CPU oracle (oracle/nmgp_oracle.py) against every golden vector generated from the reference
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle."""
import numpy as np
import pytest

from conftest import (SEP_KEYS, STA_KEYS, SVC_KEYS, golden, golden_names, hyper_dict, prior_component_err_on_the_logdet_scale, relerr,
                      vec_relerr)
from oracle import nmgp_oracle as O

VAL_TOL = 1e-6      # north-star tolerance on the log-posterior (relative); the GP-prior terms are ill-conditioned
LIK_TOL = 1e-10     # the likelihood term itself (well conditioned) agrees far tighter
GRAD_TOL = 1e-5     # analytic adjoint vs reference autograd, ||dg||/||g|| (dominated by the prior solves)


def test_primitives():
    g = golden("prims")
    assert np.allclose(O.pairwise_distances(g["X1"], g["X2"]), g["pd_12"], rtol=1e-13, atol=1e-13)
    assert np.allclose(O.pairwise_distances(g["X1"]), g["pd_11"], rtol=1e-13, atol=1e-13)
    assert np.allclose(O.RBF_cov(g["x1"], alpha=1.7, beta=0.4), g["rbf_11"], rtol=1e-13, atol=0)
    assert np.allclose(O.RBF_cov(g["x1"], g["x2"], alpha=1.7, beta=0.4), g["rbf_12"], rtol=1e-13, atol=0)
    assert np.allclose(O.RBF_cov(g["X1"], g["X2"], alpha=0.9, beta=1.3), g["rbf2d_12"], rtol=1e-13, atol=0)
    assert np.allclose(O.Nonstationary_RBF_cov(g["x1"], g["s1"], g["l1"]), g["ns_11"], rtol=1e-13, atol=0)
    assert np.allclose(O.Nonstationary_RBF_cov(g["x1"]), g["ns_11_default"], rtol=1e-13, atol=0)
    assert np.allclose(O.Nonstationary_RBF_cov(g["x1"], g["s1"], g["l1"], g["x2"], g["s2"], g["l2"]), g["ns_12"],
                       rtol=1e-13, atol=0)
    assert np.array_equal(O.kronecker_product(g["B"], g["K"]), g["kron_BK"])
    assert np.array_equal(O.kronecker_product(g["Br"], g["Kr"]), g["kron_rect"])
    assert np.array_equal(O.kronecker_product_diag(g["s1"], g["l2"]), g["kron_diag"])
    assert np.allclose(O.kron_mv(g["B"], g["K"], g["yk"]), g["kron_mv_sq"], rtol=1e-12, atol=1e-12)
    assert np.allclose(O.kron_mv(g["Br"], g["Kr"], g["yr"]), g["kron_mv_rect"], rtol=1e-12, atol=1e-12)
    assert np.allclose(O.kron_inv(float(g["sig2"]), g["B"], g["K"]), g["kron_inv"], rtol=1e-9, atol=1e-11)
    assert abs(O.kron_logdet(float(g["sig2"]), g["B"], g["K"]) - float(g["kron_logdet"])) < 1e-10
    z = np.zeros(18)
    l0 = O.multivariate_normal_logpdf0(g["yk"], z, g["B"], g["K"], float(g["sig2"]))
    l2 = O.multivariate_normal_logpdf2(g["yk"], z, g["B"], g["K"], float(g["sig2"]))
    assert relerr(l0, g["logpdf0"]) < 1e-10 and relerr(l2, g["logpdf2"]) < 1e-10
    assert relerr(l0, g["logpdf"]) < 1e-10        # reference smoke identity: logpdf0 == logpdf(kron_inv, kron_logdet)
    assert relerr(O.inverse_gamma_logpdf(0.3, 2.0, 0.7), g["invgamma"]) < 1e-14
    assert relerr(O.inverse_gamma_logpdf_u(0.3, 2.0, 0.7), g["invgamma_u"]) < 1e-14
    assert relerr(O.gamma_logpdf(0.3, 2.0, 0.7), g["gamma"]) < 1e-14
    assert np.allclose(O.uLvec2Lvec(np.arange(6) * 0.1 - 0.2, 3), g["uL2L"], rtol=1e-15)
    assert np.allclose(O.Lvec2uLvec(np.arange(1, 7) * 0.5, 3), g["L2uL"], rtol=1e-15)
    assert np.allclose(O.uLvecs2Lvecs(np.arange(12) * 0.1 - 0.5, 2, 3), g["uLs2Ls"], rtol=1e-15)
    assert np.array_equal(O.vec2lowtriangle(np.arange(1, 7), 3), g["v2tril"])
    assert np.array_equal(O.lowtriangle2vec(np.arange(9).reshape(3, 3), 3), g["tril2v"])
    # identities from the reference's own smoke blocks
    assert np.array_equal(O.uLvec2Lvec(np.zeros(6), 3), [1, 0, 1, 0, 0, 1])
    assert np.array_equal(O.vec2lowtriangle(np.arange(1, 7), 3), [[1, 0, 0], [2, 3, 0], [4, 5, 6]])
    assert np.allclose(O.kron_mv(g["Br"], g["Kr"], g["yr"]), O.kronecker_product(g["Br"], g["Kr"]) @ g["yr"])


@pytest.mark.parametrize("name", golden_names("svc_"))
def test_svc(name):
    g = golden(name)
    N, M = g["Y"].shape
    h = hyper_dict(g["hyper"], SVC_KEYS)
    prior = bool(g["prior"])
    if "Sigma" in g:
        tl, uL, tse = O.vec2pars_SVC(g["pars"], N, M)
        S = O.svc_covariance(tl, uL, tse, g["x"], M)
        assert np.allclose(S, g["Sigma"], rtol=1e-13, atol=1e-15)
        assert np.allclose(O.Nonstationary_RBF_cov(g["x"].reshape(-1, 1), ell1=np.exp(tl)), g["Kx"], rtol=1e-13, atol=0)
    big = N * M > 4000
    forms = ["cholesky"] if big else ["cholesky", "reference"]
    for form in forms:
        want_grad = form == "cholesky"
        r = O.nlogpos_obj_SVC(g["pars"], g["Y"], g["x"], **h, verbose=True, Prior=prior, formulation=form,
                              grad=want_grad)
        if want_grad:
            r, grad = r
        assert relerr(r, g["out"]) < VAL_TOL, (form, r, g["out"])
        if want_grad:
            assert vec_relerr(grad, g["grad"]) < GRAD_TOL
            assert abs(np.linalg.norm(grad) / float(g["grad_norm"]) - 1) < GRAD_TOL


def test_logpdf1_seeded_jitter():
    """a10: the reference's jittered fallback density on seeded draws (distributions.py:55-96); the stored draws also pin
    that torch's RNG stream still produces what the fixture was generated with."""
    import torch
    g = golden("prims_logpdf1")
    for k in range(int(g["ncases"])):
        torch.manual_seed(int(g["seed%d" % k]))
        jB = torch.rand(g["B%d" % k].shape[0]).type(torch.DoubleTensor).numpy()
        jK = torch.rand(g["K%d" % k].shape[0]).type(torch.DoubleTensor).numpy()
        assert np.array_equal(jB, g["jitterB%d" % k]) and np.array_equal(jK, g["jitterK%d" % k])
        y = g["y%d" % k]
        v = O.multivariate_normal_logpdf1(y, np.zeros_like(y), g["B%d" % k], g["K%d" % k], float(g["sig2_%d" % k]), jB, jK)
        assert relerr(v, g["logpdf1_%d" % k]) < 1e-9, (k, v, g["logpdf1_%d" % k])
        v0 = O.multivariate_normal_logpdf0(y, np.zeros_like(y), g["B%d" % k], g["K%d" % k], float(g["sig2_%d" % k]))
        assert relerr(v0, g["logpdf0_%d" % k]) < 1e-9
        assert v != v0                       # the jitter does move the value (it is not a no-op in the fixture)


def test_config4_subjects_at_full_size():
    """BASELINE config 4's per-GPU shape: 8 subjects, N = 1024, D = 3, mpisim hyper-parameters; reference values and
    autograd gradients at the benchmark's evaluation point and at the generating parameters."""
    g = golden("cfg4_subjects_N1024_M3")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    for s in range(g["xs"].shape[0]):
        for pk, ok, gk in (("pars", "out", "grad"), ("pars_true", "out_true", "grad_true")):
            if pk == "pars_true" and s % 4:
                continue                   # every subject at the benchmark point, two of them also at the truth
            r, grad = O.nlogpos_obj_SVC(g[pk][s], g["Ys"][s], g["xs"][s], **h, verbose=True, formulation="cholesky", grad=True)
            assert relerr(r[0], g[ok][s][0]) < VAL_TOL, (s, pk, r, g[ok][s])
            assert relerr(r[1], g[ok][s][1]) < 1e-9 and relerr(r[4], g[ok][s][4]) < 1e-12
            assert prior_component_err_on_the_logdet_scale(r[2:4], g[ok][s][2:4], g["xs"].shape[1]) < VAL_TOL
            assert vec_relerr(grad, g[gk][s]) < GRAD_TOL


@pytest.mark.parametrize("name", golden_names("sep_"))
def test_sep(name):
    g = golden(name)
    h = hyper_dict(g["hyper"], SEP_KEYS)
    r, grad = O.nlogpos_obj(g["pars"], g["Y"], g["x"], **h, verbose=True, Prior=bool(g["prior"]), grad=True)
    assert relerr(r, g["out"]) < VAL_TOL, (r, g["out"])
    assert relerr(r[1], g["out"][1]) < 1e-9            # likelihood (eigen-trick) itself
    assert relerr(r[4], g["out"][4]) < 1e-13           # Normal(0,c) prior incl. the float32 log(c) quirk
    # the reference differentiates through eigh (1/(w_i-w_j) terms at the jitter floor): its own gradient
    # carries that noise, the analytic adjoint does not -> looser bound
    assert vec_relerr(grad, g["grad"]) < 1e-4


@pytest.mark.parametrize("name", golden_names("sta_"))
def test_sta(name):
    g = golden(name)
    h = hyper_dict(g["hyper"], STA_KEYS)
    r, grad = O.nlogpos_obj_S(g["pars"], g["Y"], g["x"], **h, verbose=True, grad=True)
    assert relerr(r, g["out"]) < 1e-9, (r, g["out"])
    assert vec_relerr(grad, g["grad"]) < 1e-4


def test_prediction():
    g = golden("pred_N64_M3")
    N, M = g["Y"].shape
    h = hyper_dict(g["svc_hyper"], SVC_KEYS)
    tl, uL, tse = O.vec2pars_SVC(g["svc_pars"], N, M)
    pct, Ls, mean, var = O.predmap_inhomogeneous(tl, uL, tse, g["Y"], g["x"], g["xs"], h["mu_tilde_l"],
                                                 h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_L"], h["alpha_L"],
                                                 h["beta_L"])
    ref = g["svc_pct"]
    ref_var = ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)       # north-star tolerance: mean/var within 1e-5
    assert np.allclose(var, ref_var, rtol=1e-5, atol=1e-9)
    assert np.allclose(Ls, g["svc_Lstar"], rtol=1e-6, atol=1e-9)
    h = hyper_dict(g["sep_hyper"], SEP_KEYS)
    tl, ts, uLv, tse = O.vec2pars(g["sep_pars"], N, M)
    pct, mean, var = O.predmap_separable(tl, ts, uLv, tse, g["Y"], g["x"], g["xs"], h["mu_tilde_l"],
                                         h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_tilde_sigma"],
                                         h["alpha_tilde_sigma"], h["beta_tilde_sigma"])
    ref = g["sep_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    tl, ts, uLv, tse = O.vec2pars_S(g["sta_pars"], M)
    mean, var = O.predmap_stationary(tl, ts, uLv, tse, g["Y"], g["x"], g["xs"])
    assert np.allclose(mean, g["sta_mean"], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, g["sta_std"] ** 2, rtol=1e-5, atol=1e-9)
    assert np.allclose(mean, g["sta_pct"][:, 1], rtol=1e-5, atol=1e-7)


def test_prediction_on_the_reference_grid():
    """The 201-point grid of Nonseparable_model.py:333 at N = 512, D = 3 (pointwise_predmap_inhomogeneous, pointwise_predmap,
    pointwise_predmap_S of the reference): the oracle within the north star's 1e-5 on predictive mean / variance."""
    g = golden("pred_N512_M3_grid201")
    N, M = g["Y"].shape
    xs = g["grids"]
    assert xs.shape == (201,)
    h = hyper_dict(g["svc_hyper"], SVC_KEYS)
    tl, uL, tse = O.vec2pars_SVC(g["svc_pars"], N, M)
    pct, Ls, mean, var = O.predmap_inhomogeneous(tl, uL, tse, g["Y"], g["x"], xs, h["mu_tilde_l"], h["alpha_tilde_l"],
                                                 h["beta_tilde_l"], h["mu_L"], h["alpha_L"], h["beta_L"])
    ref = g["svc_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    # L* is a GP regression through RBF(alpha=10, beta=1) + 1e-6 I at N = 512 (condition number ~1e11): the reference solves it
    # by LU (torch.solve), we by Cholesky -- both backward stable, 2e-7 apart in absolute terms on entries of order one
    assert np.allclose(Ls, g["svc_Lstar"], rtol=1e-6, atol=1e-6)
    h = hyper_dict(g["sep_hyper"], SEP_KEYS)
    tl, ts, uLv, tse = O.vec2pars(g["sep_pars"], N, M)
    pct, mean, var = O.predmap_separable(tl, ts, uLv, tse, g["sep_Y"], g["sep_x"], xs, h["mu_tilde_l"], h["alpha_tilde_l"],
                                         h["beta_tilde_l"], h["mu_tilde_sigma"], h["alpha_tilde_sigma"],
                                         h["beta_tilde_sigma"])
    ref = g["sep_pct"]
    assert np.allclose(mean, ref[:, 1], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, ((ref[:, 2] - ref[:, 1]) / 1.96) ** 2, rtol=1e-5, atol=1e-9)
    tl, ts, uLv, tse = O.vec2pars_S(g["sta_pars"], M)
    mean, var = O.predmap_stationary(tl, ts, uLv, tse, g["sta_Y"], g["sta_x"], xs)
    assert np.allclose(mean, g["sta_mean"], rtol=1e-5, atol=1e-7)
    assert np.allclose(var, g["sta_std"] ** 2, rtol=1e-5, atol=1e-9)
