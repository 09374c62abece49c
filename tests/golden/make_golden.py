"""Generate the golden input/output vectors under tests/golden/ by RUNNING THE REFERENCE.

Only works where /root/reference exists (the build container).  The reference is imported as-is; two
aliases are installed in this process only because modern torch removed APIs the reference calls
(``torch.symeig`` -> ``torch.linalg.eigh``, ``torch.solve`` -> ``torch.linalg.solve``).
The fixtures are plain data: inputs, hyper-parameters and the reference's outputs (values + autograd gradients).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only PREFIX]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

def _symeig(A, eigenvectors=False, upper=True):
    return tuple(torch.linalg.eigh(A, UPLO="U" if upper else "L"))


torch.symeig = _symeig          # the stubs torch still ships only raise "removed"
torch.solve = lambda input, A: (torch.linalg.solve(A, input), None)

from Utility import distributions, kernels, kronecker_operation, logpos, prediction, utils  # noqa: E402  (reference)

from nonstationary_multivariate_gaussian_process_amd import sim  # noqa: E402

T64 = torch.float64


def t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote %-40s %8.1f KB" % (name, os.path.getsize(path) / 1024.0), flush=True)


def hyper_vec(h, keys):
    return np.array([float(h[k]) for k in keys])


SVC_KEYS = ["mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b"]
SEP_KEYS = ["mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma", "beta_tilde_sigma",
            "a", "b", "c"]
STA_KEYS = ["mu_tilde_l", "sigma_tilde_l", "a", "b", "c"]


def run_obj(fn, pars, Y, x, hyper, prior=True, want_grad=True):
    p = t(pars).clone().requires_grad_(want_grad)
    t0 = time.time()
    out = fn(p, t(Y), t(x), **hyper, verbose=True, Prior=prior)
    vals = np.array([float(o.detach()) for o in out])
    g = None
    if want_grad:
        out[0].backward()
        g = p.grad.detach().numpy().copy()
    return vals, g, time.time() - t0


def svc_case(name, x, Y, pars, hyper, prior=True, store_sigma=False, full_grad=True):
    N, M = Y.shape
    vals, g, dt = run_obj(logpos.nlogpos_obj_SVC, pars, Y, x, hyper, prior)
    kw = dict(kind="svc", x=x, Y=Y, pars=pars, hyper=hyper_vec(hyper, SVC_KEYS), prior=int(prior), out=vals,
              grad_norm=np.linalg.norm(g), ref_seconds=dt)
    if full_grad:
        kw["grad"] = g
    else:
        idx = np.unique(np.concatenate([np.arange(0, g.shape[0], max(1, g.shape[0] // 256)), [g.shape[0] - 1]]))
        kw["grad_idx"] = idx
        kw["grad_sub"] = g[idx]
    if store_sigma:
        T = M * (M + 1) // 2
        tl, uL, tse = logpos.vec2pars_SVC(t(pars), N, M)
        Lv = utils.uLvecs2Lvecs(uL, N, M)
        Lf = [utils.vec2lowtriangle(Lv[n * T:(n + 1) * T], M) for n in range(N)]
        Kx = kernels.Nonstationary_RBF_cov(t(x).view(-1, 1), ell1=torch.exp(tl))
        Ki = logpos.generate_K_index_SVC(Lf)
        order = torch.arange(N * M).view(N, M).t().contiguous().view(-1)
        Ki = Ki[:, order][order]
        K = kronecker_operation.kronecker_product(torch.ones(M, M, dtype=T64), Kx) * Ki
        kw["Kx"] = Kx.numpy()
        kw["Sigma"] = (K + torch.exp(tse) * torch.eye(N * M, dtype=T64)).numpy()
    save(name, **kw)


def sep_case(name, x, Y, pars, hyper, prior=True):
    vals, g, dt = run_obj(logpos.nlogpos_obj, pars, Y, x, hyper, prior)
    save(name, kind="sep", x=x, Y=Y, pars=pars, hyper=hyper_vec(hyper, SEP_KEYS), prior=int(prior), out=vals, grad=g,
         grad_norm=np.linalg.norm(g), ref_seconds=dt)


def sta_case(name, x, Y, pars, hyper):
    vals, g, dt = run_obj(logpos.nlogpos_obj_S, pars, Y, x, hyper, True)
    save(name, kind="sta", x=x, Y=Y, pars=pars, hyper=hyper_vec(hyper, STA_KEYS), prior=1, out=vals, grad=g,
         grad_norm=np.linalg.norm(g), ref_seconds=dt)


def gen_svc(only):
    for (N, M, store) in [(8, 3, True), (64, 3, True), (32, 2, True), (16, 1, True), (256, 3, False)]:
        name = "svc_rngfree_N%d_M%d" % (N, M)
        if only and not name.startswith(only):
            continue
        x, Y = sim.rngfree_inputs(N, M)
        svc_case(name, x, Y, sim.rngfree_pars_svc(N, M), sim.HYPER_SVC, store_sigma=store)
    if not only or "svc_rngfree_N64_M3_noprior".startswith(only):
        x, Y = sim.rngfree_inputs(64, 3)
        svc_case("svc_rngfree_N64_M3_noprior", x, Y, sim.rngfree_pars_svc(64, 3), sim.HYPER_SVC, prior=False)
    if not only or "svc_rngfree_N1024_M3".startswith(only):
        x, Y = sim.rngfree_inputs(1024, 3)
        svc_case("svc_rngfree_N1024_M3", x, Y, sim.rngfree_pars_svc(1024, 3), sim.HYPER_SVC)
    # simulator recipe: at the truth and at a smooth perturbation; the callers' hyper-parameter sets
    for (N, M, seed, hyper, hname, phase) in [(96, 3, 0, sim.HYPER_SVC_MPISIM, "mpisim", None),
                                              (128, 3, 1, sim.HYPER_SVC_DIST, "dist", 0.3),
                                              (100, 4, 2, sim.HYPER_SVC, "base", 1.1),
                                              (77, 2, 3, sim.HYPER_SVC, "base", 2.0),
                                              (1024, 3, 2222, sim.HYPER_SVC, "base", 0.7)]:
        name = "svc_sim_N%d_M%d_%s" % (N, M, hname)
        if only and not name.startswith(only):
            continue
        d = sim.simulate_nonseparable(N, M, seed)
        pars = d["pars_true"] if phase is None else sim.perturb(d["pars_true"], 0.05, phase)
        svc_case(name, d["x"], d["Y"], pars, hyper)
    name = "svc_sim_N2048_M3_base"
    if not only or name.startswith(only):
        d = sim.simulate_nonseparable(2048, 3, 2222)
        svc_case(name, d["x"], d["Y"], sim.perturb(d["pars_true"], 0.05, 0.7), sim.HYPER_SVC, full_grad=True)


def gen_sep(only):
    for (N, M) in [(8, 3), (64, 3), (32, 2)]:
        name = "sep_rngfree_N%d_M%d" % (N, M)
        if only and not name.startswith(only):
            continue
        x, Y = sim.rngfree_inputs(N, M)
        sep_case(name, x, Y, sim.rngfree_pars_sep(N, M), sim.HYPER_SEP)
    for (N, M, seed) in [(200, 5, 5), (512, 5, 6)]:
        name = "sep_sim_N%d_M%d" % (N, M)
        if only and not name.startswith(only):
            continue
        d = sim.simulate_separable(N, M, seed)
        sep_case(name, d["x"], d["Y"], sim.perturb(d["pars_true"], 0.05, 0.4), sim.HYPER_SEP)


def gen_sta(only):
    for (N, M) in [(8, 3), (64, 3)]:
        name = "sta_rngfree_N%d_M%d" % (N, M)
        if only and not name.startswith(only):
            continue
        x, Y = sim.rngfree_inputs(N, M)
        sta_case(name, x, Y, sim.rngfree_pars_sta(M), sim.HYPER_STA)
    name = "sta_sim_N128_M2"
    if not only or name.startswith(only):
        d = sim.simulate_stationary(128, 2, 7)
        sta_case(name, d["x"], d["Y"], sim.perturb(d["pars_true"], 0.05, 0.2), sim.HYPER_STA)


def gen_prims(only):
    if only and not "prims".startswith(only):
        return
    rng = np.random.default_rng(11)
    X1 = rng.standard_normal((7, 2))
    X2 = rng.standard_normal((5, 2))
    x1 = np.sort(rng.random(9)).reshape(-1, 1)
    x2 = np.sort(rng.random(4)).reshape(-1, 1)
    s1, l1 = np.exp(0.3 * rng.standard_normal(9)), np.exp(0.3 * rng.standard_normal(9) - 1)
    s2, l2 = np.exp(0.3 * rng.standard_normal(4)), np.exp(0.3 * rng.standard_normal(4) - 1)
    B = rng.standard_normal((3, 3)); B = B @ B.T
    K = rng.standard_normal((6, 6)); K = K @ K.T
    Br = rng.standard_normal((2, 3))
    Kr = rng.standard_normal((4, 5))
    yk = rng.standard_normal(18)
    yr = rng.standard_normal(15)
    sig2 = 0.37
    out = dict(
        X1=X1, X2=X2, x1=x1, x2=x2, s1=s1, l1=l1, s2=s2, l2=l2, B=B, K=K, Br=Br, Kr=Kr, yk=yk, yr=yr, sig2=sig2,
        pd_12=kernels.pairwise_distances(t(X1), t(X2)).numpy(),
        pd_11=kernels.pairwise_distances(t(X1)).numpy(),
        rbf_11=kernels.RBF_cov(t(x1), alpha=1.7, beta=0.4).numpy(),
        rbf_12=kernels.RBF_cov(t(x1), t(x2), alpha=1.7, beta=0.4).numpy(),
        rbf2d_12=kernels.RBF_cov(t(X1), t(X2), alpha=0.9, beta=1.3).numpy(),
        ns_11=kernels.Nonstationary_RBF_cov(t(x1), sigma1=t(s1), ell1=t(l1)).numpy(),
        ns_11_default=kernels.Nonstationary_RBF_cov(t(x1)).numpy(),
        ns_12=kernels.Nonstationary_RBF_cov(t(x1), sigma1=t(s1), ell1=t(l1), X2=t(x2), sigma2=t(s2), ell2=t(l2)).numpy(),
        kron_BK=kronecker_operation.kronecker_product(t(B), t(K)).numpy(),
        kron_rect=kronecker_operation.kronecker_product(t(Br), t(Kr)).numpy(),
        kron_diag=kronecker_operation.kronecker_product_diag(t(s1), t(l2)).numpy(),
        kron_mv_sq=kronecker_operation.kron_mv(t(B), t(K), t(yk)).numpy(),
        kron_mv_rect=kronecker_operation.kron_mv(t(Br), t(Kr), t(yr)).numpy(),
        kron_inv=kronecker_operation.kron_inv(torch.tensor(sig2, dtype=T64), t(B), t(K)).numpy(),
        kron_logdet=float(kronecker_operation.kron_logdet(torch.tensor(sig2, dtype=T64), t(B), t(K))),
        logpdf0=float(distributions.multivariate_normal_logpdf0(t(yk), torch.zeros(18, dtype=T64), t(B), t(K),
                                                                torch.tensor(sig2, dtype=T64))),
        logpdf2=float(distributions.multivariate_normal_logpdf2(t(yk), torch.zeros(18, dtype=T64), t(B), t(K),
                                                                torch.tensor(sig2, dtype=T64))),
        invgamma=float(distributions.inverse_gamma_logpdf(torch.tensor(0.3, dtype=T64), alpha=2.0, beta=0.7)),
        invgamma_u=float(distributions.inverse_gamma_logpdf_u(torch.tensor(0.3, dtype=T64), alpha=2.0, beta=0.7)),
        gamma=float(distributions.gamma_logpdf(torch.tensor(0.3, dtype=T64), alpha=2.0, beta=0.7)),
        uL2L=utils.uLvec2Lvec(t(np.arange(6) * 0.1 - 0.2), 3).numpy(),
        L2uL=utils.Lvec2uLvec(t(np.arange(1, 7) * 0.5), 3).numpy(),
        uLs2Ls=utils.uLvecs2Lvecs(t(np.arange(12) * 0.1 - 0.5), 2, 3).numpy(),
        v2tril=utils.vec2lowtriangle(t(np.arange(1, 7)), 3).numpy(),
        tril2v=utils.lowtriangle2vec(t(np.arange(9).reshape(3, 3)), 3).numpy(),
    )
    inv = torch.inverse(kronecker_operation.kronecker_product(t(B), t(K)) + sig2 * torch.eye(18, dtype=T64))
    ld = torch.logdet(kronecker_operation.kronecker_product(t(B), t(K)) + sig2 * torch.eye(18, dtype=T64))
    out["logpdf"] = float(distributions.multivariate_normal_logpdf(t(yk), torch.zeros(18, dtype=T64), ld, inv))
    save("prims", **out)


def gen_pred(only):
    if only and not "pred".startswith(only):
        return
    N, M = 64, 3
    x, Y = sim.rngfree_inputs(N, M)
    xs = np.array([0.02, 0.2, 0.37, 0.5, 0.613, 0.88, 0.99])
    h = sim.HYPER_SVC
    p = sim.rngfree_pars_svc(N, M)
    tl, uL, tse = logpos.vec2pars_SVC(t(p), N, M)
    pct, Ls = prediction.test_predmap_inhomogeneous(tl, uL, tse, t(Y), t(x), t(xs), h["mu_tilde_l"], h["alpha_tilde_l"],
                                                    h["beta_tilde_l"], h["mu_L"], h["alpha_L"], h["beta_L"])
    out = dict(x=x, Y=Y, xs=xs, svc_pars=p, svc_hyper=hyper_vec(h, SVC_KEYS), svc_pct=pct.numpy(), svc_Lstar=Ls.numpy())
    h = sim.HYPER_SEP
    p = sim.rngfree_pars_sep(N, M)
    tl, ts, uLv, tse = logpos.vec2pars(t(p), N, M)
    pct = torch.stack([prediction.point_predmap(tl, ts, uLv, tse, t(Y), t(x), t(xs)[i], h["mu_tilde_l"],
                                                h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_tilde_sigma"],
                                                h["alpha_tilde_sigma"], h["beta_tilde_sigma"]) for i in range(len(xs))])
    out.update(sep_pars=p, sep_hyper=hyper_vec(h, SEP_KEYS), sep_pct=pct.numpy())
    p = sim.rngfree_pars_sta(M)
    tl, ts, uLv, tse = logpos.vec2pars_S(t(p), M)
    mean, std = prediction.test_predmap_S(tl, ts, uLv, tse, t(Y), t(x), t(xs))
    pw = prediction.pointwise_predmap_S(tl, ts, uLv, tse, t(Y), t(x), t(xs))
    out.update(sta_pars=p, sta_mean=mean.numpy(), sta_std=std.numpy(), sta_pct=pw.numpy())
    save("pred_N64_M3", **out)


def gen_pred_grid(only):
    """Prediction at the size the reference's scripts use it: the 201-point grid linspace(0, 1, 201) of Nonseparable_model.py:333
    through pointwise_predmap_inhomogeneous (prediction.py:990-1012, one MN x MN eigendecomposition per grid point: minutes
    here), and the separable / stationary counterparts (pointwise_predmap :410-430, pointwise_predmap_S :1566-1599) on the
    same grid.  N = 512, D = 3, simulator data (seed 7), parameters = a smooth perturbation of the generating ones."""
    name = "pred_N512_M3_grid201"
    if only and not name.startswith(only):
        return
    N, M = 512, 3
    T = M * (M + 1) // 2
    grids = np.linspace(0.0, 1.0, 201)
    d = sim.simulate_nonseparable(N, M, seed=7)
    x, Y = d["x"], d["Y"]
    h = sim.HYPER_SVC
    # the curves l~(x), uL_t(x) are perturbed SMOOTHLY IN x (what a MAP estimate under the GP priors looks like).  The prediction
    # regresses them through RBF(alpha = 10, beta = 1) + 1e-6 I (condition number ~1e11): a perturbation that is rough in x
    # (sim.perturb's sine over the parameter INDEX, x being random) puts weight on the 1e-6 end of that spectrum and makes
    # the result depend on the solver's rounding at the 1e-6 level -- the reference's LU against anybody's Cholesky; that case
    # is kept as svc_rough_* with the tolerance it supports
    p = d["pars_true"].copy()
    p[:N] += 0.05 * np.sin(3.0 * x + 0.4)
    p[N:N + N * T] += (0.05 * np.sin(3.0 * x[:, None] + 0.4 + np.arange(T)[None, :])).reshape(-1)
    p[-1] += 0.05
    p_rough = sim.perturb(d["pars_true"], 0.05, 0.4)
    tl, uL, tse = logpos.vec2pars_SVC(t(p), N, M)
    t0 = time.time()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # the reference prints every grid point
        pct, Ls = prediction.pointwise_predmap_inhomogeneous(tl, uL, tse, t(Y), t(x), t(grids), h["mu_tilde_l"],
                                                             h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_L"], h["alpha_L"],
                                                             h["beta_L"])
    out = dict(x=x, Y=Y, grids=grids, svc_pars=p, svc_hyper=hyper_vec(h, SVC_KEYS), svc_pct=pct.numpy(), svc_Lstar=Ls.numpy(),
               svc_ref_seconds=time.time() - t0)
    print("  nonseparable grid: %.1f s" % (time.time() - t0), flush=True)
    tl, uL, tse = logpos.vec2pars_SVC(t(p_rough), N, M)
    with contextlib.redirect_stdout(io.StringIO()):
        pct, Ls = prediction.pointwise_predmap_inhomogeneous(tl, uL, tse, t(Y), t(x), t(grids[::10]), h["mu_tilde_l"],
                                                             h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_L"], h["alpha_L"],
                                                             h["beta_L"])
    out.update(svc_rough_pars=p_rough, svc_rough_grids=grids[::10], svc_rough_pct=pct.numpy(), svc_rough_Lstar=Ls.numpy())
    ds = sim.simulate_separable(N, M, seed=7)
    h = sim.HYPER_SEP
    ps = sim.perturb(ds["pars_true"], 0.05, 0.4)
    tl, ts, uLv, tse = logpos.vec2pars(t(ps), N, M)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        pct = prediction.pointwise_predmap(tl, ts, uLv, tse, t(ds["Y"]), t(ds["x"]), t(grids), h["mu_tilde_l"],
                                           h["alpha_tilde_l"], h["beta_tilde_l"], h["mu_tilde_sigma"], h["alpha_tilde_sigma"],
                                           h["beta_tilde_sigma"])
    out.update(sep_x=ds["x"], sep_Y=ds["Y"], sep_pars=ps, sep_hyper=hyper_vec(h, SEP_KEYS), sep_pct=pct.numpy(),
               sep_ref_seconds=time.time() - t0)
    print("  separable grid: %.1f s" % (time.time() - t0), flush=True)
    dt = sim.simulate_stationary(N, M, seed=7)
    pt = dt["pars_true"].copy()
    pt[:2] += 0.05
    tl, ts, uLv, tse = logpos.vec2pars_S(t(pt), M)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        pw = prediction.pointwise_predmap_S(tl, ts, uLv, tse, t(dt["Y"]), t(dt["x"]), t(grids))
        mean, std = prediction.test_predmap_S(tl, ts, uLv, tse, t(dt["Y"]), t(dt["x"]), t(grids))
    out.update(sta_x=dt["x"], sta_Y=dt["Y"], sta_pars=pt, sta_pct=pw.numpy(), sta_mean=mean.numpy(), sta_std=std.numpy(),
               sta_ref_seconds=time.time() - t0)
    print("  stationary grid: %.1f s" % (time.time() - t0), flush=True)
    save(name, **out)


def gen_map(only):
    """MAP trajectory (Nonseparable_model.py:147-183): Adam(lr 0.2) on [tilde_l | uL_vecs | tilde_sigma2_err]."""
    if only and not "map".startswith(only):
        return
    N, M = 64, 3
    x, Y = sim.rngfree_inputs(N, M)
    p0 = sim.rngfree_pars_svc(N, M)
    T = M * (M + 1) // 2
    tilde_l = t(p0[:N]).clone().requires_grad_(True)
    uL = t(p0[N:N + N * T]).clone().requires_grad_(True)
    tse = t(p0[-1:]).clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": tilde_l, "lr": 2e-1}, {"params": [uL, tse], "lr": 2e-1}])
    steps = 100
    hist = np.zeros(steps)
    for i in range(steps):
        opt.zero_grad()
        P = torch.cat([tilde_l, uL, tse.view(1)])
        out = logpos.nlogpos_obj_SVC(P, t(Y), t(x), **sim.HYPER_SVC, verbose=True)
        out[0].backward()
        opt.step()
        hist[i] = -float(out[0].detach())
    Pend = torch.cat([tilde_l, uL, tse.view(1)]).detach().numpy()
    save("map_svc_N64_M3", x=x, Y=Y, pars0=p0, hyper=hyper_vec(sim.HYPER_SVC, SVC_KEYS), target_value_hist=hist,
         pars_end=Pend, lr=0.2, steps=steps)


def gen_cfg4(only):
    """BASELINE config 4 at its real per-GPU shape: 8 subjects (seeds 0..7, sim.py:361-363), N = 1024, D = 3, the mpisim
    hyper-parameters (Nonseparable_model_mpisim.py:311-312), parameters = the smooth perturbation bench.py's subjects workload
    evaluates at.  One file: xs [8, N], Ys [8, N, M], pars [8, P], reference outs [8, 5] and autograd grads [8, P]."""
    name = "cfg4_subjects_N1024_M3"
    if only and not name.startswith(only):
        return
    N, M, S = 1024, 3, 8
    xs, Ys, ps, outs, grads, secs, pts, outs_t, grads_t = [], [], [], [], [], [], [], [], []
    for s in range(S):
        d = sim.simulate_nonseparable(N, M, seed=s)
        pars = sim.perturb(d["pars_true"], 0.05, 0.7)
        vals, g, dt = run_obj(logpos.nlogpos_obj_SVC, pars, d["Y"], d["x"], sim.HYPER_SVC_MPISIM, True)
        # ... and at the generating parameters (smooth curves: the GP-prior terms do not swamp the likelihood there)
        vt, gt, dt2 = run_obj(logpos.nlogpos_obj_SVC, d["pars_true"], d["Y"], d["x"], sim.HYPER_SVC_MPISIM, True)
        xs.append(d["x"]); Ys.append(d["Y"]); ps.append(pars); outs.append(vals); grads.append(g); secs.append(dt)
        pts.append(d["pars_true"]); outs_t.append(vt); grads_t.append(gt)
        print("  subject %d: NegLog %.10g (%.1f s), at truth %.10g" % (s, vals[0], dt, vt[0]), flush=True)
    save(name, kind="cfg4", xs=np.stack(xs), Ys=np.stack(Ys), pars=np.stack(ps), out=np.stack(outs), grad=np.stack(grads),
         pars_true=np.stack(pts), out_true=np.stack(outs_t), grad_true=np.stack(grads_t),
         hyper=hyper_vec(sim.HYPER_SVC_MPISIM, SVC_KEYS), seeds=np.arange(S), ref_seconds=np.array(secs))


def gen_sep_big(only):
    """BASELINE config 5 at its real shape: separable model, N = 4096, D = 5 (logpos.py:216-296)."""
    name = "sep_sim_N4096_M5"
    if only and not name.startswith(only):
        return
    d = sim.simulate_separable(4096, 5, 8)
    sep_case(name, d["x"], d["Y"], sim.perturb(d["pars_true"], 0.05, 0.4), sim.HYPER_SEP)


def gen_logpdf1(only):
    """multivariate_normal_logpdf1 (distributions.py:55-96): the jitter is drawn with torch.rand (B first, then K), so a
    torch.manual_seed right before the call makes the reference's value reproducible."""
    name = "prims_logpdf1"
    if only and not name.startswith(only):
        return
    rng = np.random.default_rng(21)
    out = {}
    cases = []
    # (a) generic SPD B, K; (b) K with exactly repeated eigenvalues (the situation the jitter exists for: RBF rows that
    # coincide); (c) a smooth Gibbs K at the 1e-6 jitter floor with a rank-deficient B
    B = rng.standard_normal((3, 3)); B = B @ B.T
    K = rng.standard_normal((6, 6)); K = K @ K.T
    cases.append((B, K, rng.standard_normal(18), 0.37, 101))
    Q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    K2 = (Q * np.array([2.0] * 4 + [0.5] * 4 + [1e-3] * 4)) @ Q.T
    K2 = 0.5 * (K2 + K2.T)
    B2 = np.array([[1.0, 0.3], [0.3, 1.0]])
    cases.append((B2, K2, rng.standard_normal(24), 0.05, 202))
    x = np.linspace(0.05, 0.95, 40)
    K3 = kernels.Nonstationary_RBF_cov(t(x).view(-1, 1), sigma1=t(np.exp(0.3 * np.sin(3 * x))),
                                       ell1=t(np.exp(3 * (x - 1) ** 3 - 1))).numpy()
    l = np.array([[1.0, 0, 0], [0.5, 1e-4, 0], [0.2, 0.1, 1e-4]])
    cases.append((l @ l.T, K3, rng.standard_normal(120), 1e-2, 303))
    for k, (Bk, Kk, yk, s2, seed) in enumerate(cases):
        torch.manual_seed(seed)
        v1 = float(distributions.multivariate_normal_logpdf1(t(yk), torch.zeros(len(yk), dtype=T64), t(Bk), t(Kk),
                                                             torch.tensor(s2, dtype=T64)))
        v0 = float(distributions.multivariate_normal_logpdf0(t(yk), torch.zeros(len(yk), dtype=T64), t(Bk), t(Kk),
                                                             torch.tensor(s2, dtype=T64)))
        torch.manual_seed(seed)
        jB = torch.rand(Bk.shape[0]).type(torch.DoubleTensor).numpy()
        jK = torch.rand(Kk.shape[0]).type(torch.DoubleTensor).numpy()
        out.update({"B%d" % k: Bk, "K%d" % k: Kk, "y%d" % k: yk, "sig2_%d" % k: s2, "seed%d" % k: seed,
                    "logpdf1_%d" % k: v1, "logpdf0_%d" % k: v0, "jitterB%d" % k: jB, "jitterK%d" % k: jK})
        print("  logpdf1 case %d: %.12g (logpdf0 %.12g)" % (k, v1, v0), flush=True)
    out["ncases"] = len(cases)
    save(name, **out)


def gen_local_estimation(only):
    """empirical_estimation.local_estimation (empirical_estimation.py:71-133) on one simulated subject, two window sizes."""
    name = "init_local_estimation_N90_M2"
    if only and not name.startswith(only):
        return
    from Utility import empirical_estimation as EE
    d = sim.simulate_nonseparable(90, 2, seed=4)
    out = dict(x=d["x"], Y=d["Y"])
    for W in (12, 30):
        t0 = time.time()
        es, el, sl, st, R, B, Lv, tse = EE.local_estimation(d["x"], d["Y"], window_size=W)
        out.update({"w%d_sigmas" % W: es, "w%d_ls" % W: el, "w%d_smooth_ls" % W: sl, "w%d_stds" % W: st, "w%d_R" % W: R,
                    "w%d_B" % W: B, "w%d_L_vecs" % W: Lv, "w%d_tse" % W: float(tse)})
        print("  local_estimation W=%d: %.1f s" % (W, time.time() - t0), flush=True)
    S, Lg = EE.global_estimation(d["x"], d["Y"])
    out.update(global_S=S, global_L_vec=Lg)
    save(name, **out)


def gen_map_sep_sta(only):
    """MAP trajectories of the separable (Separable_model.py:147-166: Adam, lr 0.2 on both groups) and stationary
    (Stationary_model.py:112-131: Adam lr 0.1 on [tilde_l, uL_vec, tilde_sigma2_err], tilde_sigma fixed) loops."""
    name = "map_sep_sta_N64_M3"
    if only and not name.startswith(only):
        return
    N, M = 64, 3
    T = M * (M + 1) // 2
    x, Y = sim.rngfree_inputs(N, M)
    steps = 60
    # separable
    p0 = sim.rngfree_pars_sep(N, M)
    tl = t(p0[:N]).clone().requires_grad_(True)
    ts = t(p0[N:2 * N]).clone().requires_grad_(True)
    uL = t(p0[2 * N:2 * N + T]).clone().requires_grad_(True)
    tse = t(p0[-1:]).clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [ts, uL, tse], "lr": 0.2}, {"params": tl, "lr": 0.2}])
    hist = np.zeros(steps)
    for i in range(steps):
        opt.zero_grad()
        P = torch.cat([tl, ts, uL, tse.view(1)])
        out = logpos.nlogpos_obj(P, t(Y), t(x), **sim.HYPER_SEP, verbose=True)
        out[0].backward()
        opt.step()
        hist[i] = -float(out[0].detach())
    sep_end = torch.cat([tl, ts, uL, tse.view(1)]).detach().numpy()
    # stationary
    q0 = sim.rngfree_pars_sta(M)
    tl1 = t(q0[:1]).clone().requires_grad_(True)
    ts1 = t(q0[1:2]).clone()
    uL1 = t(q0[2:2 + T]).clone().requires_grad_(True)
    tse1 = t(q0[-1:]).clone().requires_grad_(True)
    opt = torch.optim.Adam([tl1, uL1, tse1], lr=0.1)
    hist_s = np.zeros(steps)
    for i in range(steps):
        opt.zero_grad()
        P = torch.cat([tl1, ts1, uL1, tse1.view(1)])
        out = logpos.nlogpos_obj_S(P, t(Y), t(x), **sim.HYPER_STA, verbose=True)
        out[0].backward()
        opt.step()
        hist_s[i] = -float(out[0].detach())
    sta_end = torch.cat([tl1, ts1, uL1, tse1.view(1)]).detach().numpy()
    save(name, x=x, Y=Y, steps=steps, sep_pars0=p0, sep_hyper=hyper_vec(sim.HYPER_SEP, SEP_KEYS), sep_hist=hist, sep_end=sep_end,
         sta_pars0=q0, sta_hyper=hyper_vec(sim.HYPER_STA, STA_KEYS), sta_hist=hist_s, sta_end=sta_end)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    gen_prims(a.only)
    gen_svc(a.only)
    gen_sep(a.only)
    gen_sta(a.only)
    gen_pred(a.only)
    gen_pred_grid(a.only)
    gen_map(a.only)
    gen_cfg4(a.only)
    gen_sep_big(a.only)
    gen_logpdf1(a.only)
    gen_local_estimation(a.only)
    gen_map_sep_sta(a.only)
