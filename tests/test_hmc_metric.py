"""The prior-factor metric of the device-resident HMC trajectories (csrc/nmgp_metric.hip, nmgp_svc_batch_traj_set_mass_prior,
drivers.PriorMetric / prior_lowrank_metric): the replacement for the `M = inv(sample covariance)` the reference's production sampler
call passes (Nonseparable_model_mpiKAISER.py:267-270,398-411) at sizes where a sample covariance is out of reach.

The change of coordinates is checked against the oracle's GP-prior covariances (logpos.py:357-365), the trajectories against the
dense-mass path with the SAME metric written out as a [P, P] matrix, the sampler against the dense-mass sampler on posterior
summaries; B chains in one launch sequence must give the bits of B single-chain runs."""
import numpy as np
import pytest

from conftest import SVC_KEYS, golden, hyper_dict


def _blocks(x, N, T, hv):
    """Dense covariances / Cholesky factors of the two GP priors from the oracle (checker)."""
    from oracle import nmgp_oracle as O
    Kl = O.RBF_cov(x[:, None], None, hv[1], hv[2])
    KL = O.RBF_cov(x[:, None], None, hv[4], hv[5])
    return Kl, KL


def _apply_blocks(Al, AL, v, N, T):
    """blockdiag(Al, AL per stride-T column, 1) v for parameter-shaped rows v [B, P]."""
    out = np.empty_like(v)
    out[:, :N] = v[:, :N] @ Al.T
    U = v[:, N:N + N * T].reshape(-1, N, T)
    out[:, N:N + N * T] = np.einsum("ik,bkt->bit", AL, U).reshape(v.shape[0], -1)
    out[:, -1] = v[:, -1]
    return out


def _dense_Lblk(ctx, hv, P, B):
    """L_blk as a dense [P, P] matrix through the C ABI (columns = images of the unit vectors)."""
    Lb = np.zeros((P, P))
    eye = np.eye(P)
    for a in range(0, P, B):
        m = min(B, P - a)
        buf = np.zeros((B, P))
        buf[:m] = eye[a:a + m]
        Lb[:, a:a + m] = ctx.svc_batch_prior_apply(hv, buf, trans=False)[:m].T
    return Lb


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,hv", [
    (96, 3, [0.0, 10.0, 1.0, 0.0, 10.0, 1.0, 1.0, 1.0]),          # the scripts' hyper-parameters: ONE factor for both blocks
    (130, 2, [0.3, 5.0, 0.1, -0.2, 2.0, 0.2, 1.0, 1.0]),          # two different factors, N not a multiple of the 64-row tile
    (64, 4, [0.0, 1.0, 0.05, 0.0, 3.0, 0.07, 1.0, 1.0]),          # T = 10: two column groups per workgroup row
])
def test_prior_apply_is_the_cholesky_factor_of_the_gp_prior_covariances(N, M, hv):
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    d = sim.simulate_nonseparable(N, M, seed=4)
    T = M * (M + 1) // 2
    P = N * (1 + T) + 1
    B = 5
    hv = np.array(hv)
    Kl, KL = _blocks(d["x"], N, T, hv)
    v = np.random.default_rng(1).standard_normal((B, P))
    c = _lib.Context(0)
    try:
        c.set_data(d["x"], d["Y"])
        c.svc_batch_alloc(B)
        Ltv = c.svc_batch_prior_apply(hv, v, trans=True)
        LLtv = c.svc_batch_prior_apply(hv, Ltv, trans=False)
        want = _apply_blocks(Kl, KL, v, N, T)
        # L (L^T v) = Sigma v: well conditioned whatever the condition number of Sigma
        assert np.linalg.norm(LLtv - want) / np.linalg.norm(want) < 1e-11
        # ... and each orientation on its own against NumPy's factor (the factor of RBF + 1e-6 I, kappa ~ 1e9 .. 1e11 at these
        # spacings, is itself only defined to kappa eps: the bar is loose here, the identity above is the tight one)
        Ll, LL = np.linalg.cholesky(Kl), np.linalg.cholesky(KL)
        tol = 1e-5
        Lv = c.svc_batch_prior_apply(hv, v, trans=False)
        for got, ref in ((Lv, _apply_blocks(Ll, LL, v, N, T)), (Ltv, _apply_blocks(Ll.T, LL.T, v, N, T))):
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol
        # lower triangular: the image of the first unit vector of a block is the factor's first column
        e = np.zeros((B, P))
        e[0, 0] = 1.0
        e[1, N + 2] = 1.0                   # location 0, uL column 2
        col = c.svc_batch_prior_apply(hv, e, trans=False)
        assert np.allclose(col[0, :N], Ll[:, 0], rtol=1e-10, atol=1e-13) and np.all(col[0, N:] == 0)
        assert np.allclose(col[1, N + 2:N + N * T:T], LL[:, 0], rtol=1e-10, atol=1e-13) and np.all(col[1, :N] == 0)
        assert col[1, -1] == 0 and np.count_nonzero(col[1]) <= N
    finally:
        c.close()


@pytest.mark.gpu
def test_prior_apply_uses_each_subjects_own_factors():
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    N, M, S, k = 72, 2, 3, 2
    T = 3
    P = N * (1 + T) + 1
    hv = np.array([0.0, 4.0, 0.15, 0.0, 2.0, 0.1, 1.0, 1.0])
    ds = [sim.simulate_nonseparable(N, M, seed=10 + s) for s in range(S)]
    xs, Ys = np.stack([d["x"] for d in ds]), np.stack([d["Y"] for d in ds])
    v = np.random.default_rng(2).standard_normal((S * k, P))
    c = _lib.Context(0)
    try:
        c.set_data(xs[0], Ys[0])
        c.svc_batch_alloc(S * k)
        c.svc_batch_set_subjects(xs, Ys, k)
        got = c.svc_batch_prior_apply(hv, c.svc_batch_prior_apply(hv, v, trans=True), trans=False)
        for s in range(S):
            Kl, KL = _blocks(xs[s], N, T, hv)
            want = _apply_blocks(Kl, KL, v[s * k:(s + 1) * k], N, T)
            assert np.linalg.norm(got[s * k:(s + 1) * k] - want) / np.linalg.norm(want) < 1e-11, s
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,hv", [
    (96, 3, [0.0, 10.0, 1.0, 0.0, 10.0, 1.0, 1.0, 1.0, 10.0]),     # the scripts' hyper-parameters: one factor for both GP blocks
    (130, 2, [0.3, 5.0, 0.1, -0.2, 2.0, 0.2, 1.0, 1.0, 2.3025851]),   # two factors, ragged N, a c that float32 rounds
])
def test_separable_prior_apply_is_the_block_factor_of_the_separable_priors(N, M, hv):
    """nmgp_sep_prior_apply: L_blk = blockdiag(chol Sigma_l, chol Sigma_sigma, c I_T, 1) on [tilde_l | tilde_sigma | uL_vec | s2]."""
    from oracle import nmgp_oracle as O
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    d = sim.simulate_separable(N, M, 4)
    T = M * (M + 1) // 2
    P = 2 * N + T + 1
    hv = np.array(hv)
    Kl = O.RBF_cov(d["x"][:, None], None, hv[1], hv[2])
    Ks = O.RBF_cov(d["x"][:, None], None, hv[4], hv[5])
    v = np.random.default_rng(3).standard_normal((4, P))
    c = _lib.Context(0)
    try:
        c.set_data(d["x"], d["Y"])
        Ltv = c.sep_prior_apply(hv, v, trans=True)
        got = c.sep_prior_apply(hv, Ltv, trans=False)
        c32 = float(np.float32(hv[8]))
        want = np.concatenate([v[:, :N] @ Kl.T, v[:, N:2 * N] @ Ks.T, c32 * c32 * v[:, 2 * N:2 * N + T], v[:, -1:]], axis=1)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-11
        assert np.array_equal(Ltv[:, 2 * N:2 * N + T], c32 * v[:, 2 * N:2 * N + T]) and np.array_equal(Ltv[:, -1], v[:, -1])
        e = np.zeros((2, P))
        e[0, 0] = 1.0
        e[1, N] = 1.0
        col = c.sep_prior_apply(hv, e, trans=False)
        Ll, Ls = np.linalg.cholesky(Kl), np.linalg.cholesky(Ks)
        assert np.allclose(col[0, :N], Ll[:, 0], rtol=1e-10, atol=1e-13) and np.all(col[0, N:] == 0)
        assert np.allclose(col[1, N:2 * N], Ls[:, 0], rtol=1e-10, atol=1e-13) and np.all(col[1, :N] == 0) and np.all(col[1, 2 * N:] == 0)
    finally:
        c.close()


def _metric_pieces(c, hv, P, B, r, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((P, r)))
    U = np.ascontiguousarray(Q.T)
    lam = np.exp(rng.uniform(np.log(0.5), np.log(500.0), r))
    Lb = _dense_Lblk(c, hv, P, B)
    W = np.eye(P) - U.T @ np.diag(lam / (1 + lam)) @ U                   # (I + U lam U^T)^-1
    Rw = np.eye(P) + U.T @ np.diag(np.sqrt(1 + lam) - 1) @ U             # (I + U lam U^T)^1/2
    return U, lam, Lb, W, Rw


@pytest.mark.gpu
def test_prior_metric_trajectories_equal_the_dense_mass_path_with_the_same_matrix():
    """kind 3 against kind 2: M^-1 = L_blk W L_blk^T written out, momenta p0 = L_blk^-T W^-1/2 z handed to nmgp_svc_batch_traj."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    g = golden("svc_rngfree_N32_M2")
    hv = g["hyper"].copy()
    hv[2] = hv[5] = 0.1                       # well-conditioned prior factors: the two paths agree to rounding
    N, M = g["Y"].shape
    B, r, nsteps = 3, 5, 7
    q = np.stack([sim.perturb(g["pars"], 0.01 * (b + 1), 0.4 * b) for b in range(B)])
    P = q.shape[1]
    c = _lib.Context(0)
    try:
        c.set_data(g["x"], g["Y"])
        c.svc_batch_alloc(B)
        U, lam, Lb, W, Rw = _metric_pieces(c, hv, P, B, r, 3)
        Minv = Lb @ W @ Lb.T
        z = np.random.default_rng(5).standard_normal((B, P))
        p0 = np.linalg.solve(Lb.T, (z @ Rw.T).T).T                       # p = L^-T u, u = W^-1/2 z

        def start():
            c.svc_batch_set_pars(q)
            c.svc_batch_eval(hv, True, want_grad=True)
            out, st = c.svc_batch_fetch()
            assert (st == 0).all()
            return out[:, 0].copy()
        for eps in (0.004, 0.002):
            U0 = start()
            c.svc_batch_traj_set_mass(Minv)
            c.svc_batch_traj_begin()
            qd, pd, Ud, fd = c.svc_batch_traj(hv, True, eps, nsteps, p0)
            kd = 0.5 * np.einsum("bi,ij,bj->b", pd, Minv, pd)
            U0b = start()
            assert np.array_equal(U0, U0b)
            c.svc_batch_traj_set_mass_prior(hv, U, lam)
            c.svc_batch_traj_begin()
            with pytest.raises(_lib.NmgpError, match="traj_z"):
                c.svc_batch_traj(hv, True, eps, nsteps, p0)              # whitened momenta: only the _z entry runs
            qp, kp, Up, fp = c.svc_batch_traj_z(hv, True, eps, nsteps, z)
            assert not fd.any() and not fp.any()
            assert np.allclose(qp, qd, rtol=1e-9, atol=1e-11)
            # (the host-side reference 1/2 p^T M^-1 p goes through L_blk^-T and the written-out M^-1, condition ~1e9: 1e-6)
            assert np.allclose(Up, Ud, rtol=1e-9) and np.allclose(kp, kd, rtol=1e-6)
            assert np.abs(qp - q).max() > 1e-3                           # the chains went somewhere
            dH = (Up + kp) - (U0 + 0.5 * (z * z).sum(1))
            assert np.abs(dH).max() < 1.0
            if eps == 0.004:
                dH_big = dH
            else:
                # a second-order integrator: halving the step divides the energy error by ~4
                assert np.abs(dH).max() < 0.4 * np.abs(dH_big).max()
            # rejected chains get their state back, accepted ones keep it
            c.svc_batch_traj_commit(np.array([1, 0, 1], dtype=bool))
            now = c.svc_batch_get_pars()
            assert np.array_equal(now[1], q[1]) and np.array_equal(now[0], qp[0]) and np.array_equal(now[2], qp[2])
        # rank 0 = the pure prior metric
        start()
        c.svc_batch_traj_set_mass(Lb @ Lb.T)
        c.svc_batch_traj_begin()
        qd, pd, Ud, fd = c.svc_batch_traj(hv, True, 0.005, 3, np.linalg.solve(Lb.T, z.T).T)
        start()
        c.svc_batch_traj_set_mass_prior(hv)
        c.svc_batch_traj_begin()
        qp, kp, Up, fp = c.svc_batch_traj_z(hv, True, 0.005, 3, z)
        assert np.allclose(qp, qd, rtol=1e-9, atol=1e-11) and np.allclose(kp, 0.5 * np.einsum("bi,ij,bj->b", pd, Lb @ Lb.T, pd), rtol=1e-6)
    finally:
        c.close()


@pytest.mark.gpu
def test_prior_metric_chains_in_one_batch_give_the_bits_of_single_chain_runs():
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, PriorMetric
    g = golden("svc_rngfree_N64_M3")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    B, S, L = 4, 3, 5
    init = np.stack([sim.perturb(g["pars"], 0.005 * (b + 1), 0.3 * b) for b in range(B)])
    P = init.shape[1]
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((P, 6)))
    met = PriorMetric(h, np.ascontiguousarray(Q.T), np.array([300.0, 90.0, 20.0, 5.0, 2.0, 0.7]))
    hb = BatchedHMC(g["x"], g["Y"], h, init, step_size=0.002, num_steps_in_leap=L, seed=21, M=met)
    sb, ib = hb.run(S)
    assert np.abs(sb[-1] - init).max() > 1e-4 and np.all(ib["accept_rate"] > 0)
    assert 0.0 < ib["timing"]["device_share"] <= 1.0
    for b in range(B):
        h1 = BatchedHMC(g["x"], g["Y"], h, init[b:b + 1], step_size=0.002, num_steps_in_leap=L, seed=21 + b, M=met)
        s1, i1 = h1.run(S)
        assert np.array_equal(s1[:, 0], sb[:, b]), b
        assert np.array_equal(i1["energy_error"][:, 0], ib["energy_error"][:, b], equal_nan=True)
    with pytest.raises(ValueError, match="device_resident"):
        BatchedHMC(g["x"], g["Y"], h, init, M=met, device_resident=False)


@pytest.mark.gpu
def test_prior_metric_of_a_multi_subject_batch_uses_each_subjects_factors_and_directions():
    """S subjects x k chains in one batch (config 4's unit) under per-subject metrics (PriorMetric.stack: U [S, r, P], ranks padded):
    every subject's chains must give the bits of that subject sampled on its own with its own metric."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, PriorMetric
    N, M, S, k, L, ns = 40, 2, 3, 2, 4, 3
    h = dict(sim.HYPER_SVC)
    ds = [sim.simulate_nonseparable(N, M, seed=30 + s) for s in range(S)]
    xs, Ys = np.stack([d["x"] for d in ds]), np.stack([d["Y"] for d in ds])
    P = N * 4 + 1
    rng = np.random.default_rng(4)
    mets = []
    for s_ in range(S):
        r = 2 + s_                                            # different ranks: the stack pads
        Q, _ = np.linalg.qr(rng.standard_normal((P, r)))
        mets.append(PriorMetric(h, np.ascontiguousarray(Q.T), np.exp(rng.uniform(0.0, 5.0, r))))
    met = PriorMetric.stack(mets)
    assert met.U.shape == (S, S + 1, P) and met.lam[0, 2:].max() == 0.0
    init = np.stack([sim.perturb(ds[s_]["pars_true"], 0.01 * (c + 1), 0.2 * c) for s_ in range(S) for c in range(k)])
    hb = BatchedHMC(xs, Ys, h, init, step_size=0.002, num_steps_in_leap=L, seed=50, M=met, chains_per_subject=k)
    sb, ib = hb.run(ns)
    assert np.abs(sb[-1] - init).max() > 1e-4
    for s_ in range(S):
        h1 = BatchedHMC(ds[s_]["x"], ds[s_]["Y"], h, init[s_ * k:(s_ + 1) * k], step_size=0.002, num_steps_in_leap=L, seed=50 + s_ * k,
                        M=mets[s_])
        s1, i1 = h1.run(ns)
        assert np.array_equal(s1, sb[:, s_ * k:(s_ + 1) * k]), s_
        assert np.array_equal(i1["energy_error"], ib["energy_error"][:, s_ * k:(s_ + 1) * k], equal_nan=True)


@pytest.mark.gpu
def test_lowrank_metric_finds_the_likelihood_curvature_in_whitened_coordinates():
    """prior_lowrank_metric against the dense whitened Hessian of the ORACLE's likelihood gradient (central differences)."""
    from oracle import nmgp_oracle as O
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import prior_lowrank_metric
    N, M = 40, 2
    T = 3
    d = sim.simulate_nonseparable(N, M, seed=6)
    h = dict(sim.HYPER_SVC)
    hv = np.array([h[k] for k in SVC_KEYS])
    q = sim.perturb(d["pars_true"], 0.02, 0.2)
    P = q.shape[0]
    met = prior_lowrank_metric(d["x"], d["Y"], h, q, rank=24, oversample=16, power_iters=2, h=1e-3, seed=1, batch=8)
    assert met.info["grad_evals"] == 4 * 2 * 40
    # reference: A = L^T H L column by column
    Kl, KL = _blocks(d["x"], N, T, hv)
    Ll, LL = np.linalg.cholesky(Kl), np.linalg.cholesky(KL)

    def lik_grad(p):
        return O.nlogpos_obj_SVC(p, d["Y"], d["x"], **h, verbose=True, grad=True, Prior=False)[1]
    A = np.zeros((P, P))
    eye = np.eye(P)
    Lcols = _apply_blocks(Ll, LL, eye, N, T)            # row k = L_blk e_k
    for k in range(P):
        dq = Lcols[k]
        t = 1e-3 / np.abs(dq).max()
        A[:, k] = _apply_blocks(Ll.T, LL.T, ((lik_grad(q + t * dq) - lik_grad(q - t * dq)) / (2 * t))[None], N, T)[0]
    ev, V = np.linalg.eigh(0.5 * (A + A.T))
    order = np.argsort(-np.abs(ev))                     # the metric keeps the largest |eigenvalues| (negative ones as |lam|)
    ev, V = ev[order], V[:, order]
    r = met.rank
    assert r >= 8 and ev[0] > 100.0
    big = met.lam > 5.0
    assert np.allclose(met.lam[big], np.abs(ev[:r])[big], rtol=0.02)
    # the subspace of the well-separated leading eigenvalues
    k = int(np.sum(np.abs(ev) > 20.0))
    overlap = np.linalg.svd(met.U[:k] @ V[:, :k], compute_uv=False)
    assert overlap.min() > 0.98
    assert np.allclose(met.U @ met.U.T, np.eye(r), atol=1e-10)


@pytest.mark.gpu
def test_prior_metric_sampler_agrees_with_the_dense_mass_sampler_and_mixes():
    """Same posterior, two implementations of the same metric: posterior summaries agree within Monte-Carlo error, and -- unlike
    the identity mass at the same cost -- the chains mix (split R-hat).  The chains start from a polished MAP estimate, as the
    N = 2048 run does (tools/hmc_1000.py)."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, BatchedMAP, polish_map, prior_lowrank_metric
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hmc_1000 as H
    N, M = 64, 3
    T = 6
    d = sim.simulate_nonseparable(N, M, seed=5)
    h = dict(sim.HYPER_SVC)
    hv = np.array([h[k] for k in SVC_KEYS])
    p0 = sim.perturb(d["pars_true"], 0.01, 0.1)
    pars, hist, alive = BatchedMAP(d["x"][None], d["Y"][None], h, p0[None], lr=0.02).run(400)      # the reference's optimiser ...
    q0, nl, gn, nev = polish_map(d["x"], d["Y"], h, pars[0], maxiter=300, probes=40, rank=32)                          # ... then to the mode
    assert -nl >= hist[-1, 0] - 1e-6
    P = q0.shape[0]
    B, S, L, eps = 8, 400, 10, 0.15
    met = prior_lowrank_metric(d["x"], d["Y"], h, q0, rank=32, oversample=8, power_iters=1, seed=3, batch=B)
    assert met.info["most_negative"] > -0.9          # at a mode the whitened Hessian I + A is positive definite
    init = np.repeat(q0[None], B, 0)
    hp = BatchedHMC(d["x"], d["Y"], h, init, step_size=eps, num_steps_in_leap=L, seed=40, M=met)
    sp, ip = hp.run(S)
    assert ip["accept_rate"].mean() > 0.7
    # the same metric as a dense matrix through the kind-2 path
    c = _lib.default_context()
    Lb = _dense_Lblk(c, hv, P, B)
    W = np.eye(P) - met.U.T @ np.diag(met.lam / (1 + met.lam)) @ met.U
    Minv = Lb @ W @ Lb.T
    Minv = 0.5 * (Minv + Minv.T)
    hd = BatchedHMC(d["x"], d["Y"], h, init, step_size=eps, num_steps_in_leap=L, seed=77, Minv=Minv)
    sdn, idn = hd.run(S)
    assert abs(idn["accept_rate"].mean() - ip["accept_rate"].mean()) < 0.1
    half = S // 2
    a, b = sp[half:], sdn[half:]
    rh = H.split_rhat(a)
    assert np.nanmedian(rh) < 1.05 and np.nanquantile(rh, 0.99) < 1.3
    # summaries: log sigma^2, tilde_l at three locations, one uL entry
    idx = [P - 1, 0, N // 2, N - 1, N + 3 * T + 1]
    for i in idx:
        ma, mb = a[:, :, i].mean(), b[:, :, i].mean()
        sd = 0.5 * (a[:, :, i].std() + b[:, :, i].std())
        ess = min(H.multichain_ess(a[:, :, [i]])[0], H.multichain_ess(b[:, :, [i]])[0])
        assert abs(ma - mb) < 5.0 * sd * np.sqrt(2.0 / max(ess, 10.0)), (i, ma, mb, sd, ess)
        assert 0.6 < a[:, :, i].std() / b[:, :, i].std() < 1.6
    # identity mass with the same number of gradient evaluations does not get there (what the metric is for)
    hi = BatchedHMC(d["x"], d["Y"], h, init, step_size=2e-4, num_steps_in_leap=L, seed=40)
    si, ii = hi.run(S)
    assert np.nanmedian(H.split_rhat(si[half:])) > 1.5


@pytest.mark.gpu
def test_metric_preconditioned_lbfgs_reaches_the_mode_where_adam_does_not():
    """drivers.polish_map from the START point (no Adam): L-BFGS in the prior-whitened coordinates with the low-rank likelihood
    curvature as initial matrix.  It must end at a stationary point (whitened gradient ~0, positive definite whitened Hessian) whose
    log posterior is at least what the reference's Adam recipe (Nonseparable_model.py:147-210) reaches in 600 iterations from the
    same start, and both routes -- from the start, and from Adam's end point -- must find the same mode."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedMAP, polish_map, prior_lowrank_metric
    N, M = 64, 3
    d = sim.simulate_nonseparable(N, M, seed=5)
    h = dict(sim.HYPER_SVC)
    p0 = sim.perturb(d["pars_true"], 0.05, 0.7)
    pars, hist, alive = BatchedMAP(d["x"][None], d["Y"][None], h, p0[None], lr=0.2).run(600)
    qa, nla, gna, neva = polish_map(d["x"], d["Y"], h, p0, maxiter=400, rounds=8, probes=40, rank=32)
    qb, nlb, gnb, nevb = polish_map(d["x"], d["Y"], h, pars[0], maxiter=300, probes=40, rank=32)
    assert -nla >= hist[-1, 0] and -nla >= hist.max() - 1e-6              # at least as high as anything Adam visited
    assert abs(nla - nlb) < 1e-6 * abs(nla) and np.sqrt(np.mean((qa - qb) ** 2)) < 1e-2
    assert gna < 1e-5 * abs(nla) + 1e-3 and gnb < 1e-5 * abs(nlb) + 1e-3
    met = prior_lowrank_metric(d["x"], d["Y"], h, qa, rank=32, oversample=8, seed=2, batch=8)
    assert met.info["most_negative"] > -0.9                               # I + A positive definite: a mode, not a saddle


def test_separable_prior_metric_algebra_cpu():
    """SeparablePriorMetric on the host: root(z) has covariance I + U lam U^T, W is its inverse, the kinetic energy of root(z) is
    1/2 |z|^2, and apply / apply^T are the block products with the factors they were given."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import SeparablePriorMetric
    rng = np.random.default_rng(0)
    N, T, r, B = 7, 3, 4, 5
    P = 2 * N + T + 1
    L_l = np.tril(rng.standard_normal((N, N))) + 3 * np.eye(N)
    L_s = np.tril(rng.standard_normal((N, N))) + 3 * np.eye(N)
    Q, _ = np.linalg.qr(rng.standard_normal((P, r)))
    lam = np.array([50.0, 7.0, 1.5, 0.6])
    met = SeparablePriorMetric(L_l, L_s, 10.0, T, Q.T, lam)
    Lb = np.zeros((P, P))
    Lb[:N, :N], Lb[N:2 * N, N:2 * N] = L_l, L_s
    Lb[2 * N:2 * N + T, 2 * N:2 * N + T] = 10.0 * np.eye(T)
    Lb[-1, -1] = 1.0
    v = rng.standard_normal((B, P))
    assert np.allclose(met.apply(v, False), v @ Lb.T) and np.allclose(met.apply(v, True), v @ Lb)
    H = np.eye(P) + Q @ np.diag(lam) @ Q.T
    Rm = np.stack([met.root(e) for e in np.eye(P)[None]])[0]                     # rows: root(e_k)
    assert np.allclose(Rm @ Rm.T, H) and np.allclose(met.W(v) @ H, v)
    assert np.allclose(met.kinetic(met.root(v)), 0.5 * (v * v).sum(1))
    assert np.allclose(met.kinetic(v), 0.5 * np.einsum("bi,ij,bj->b", v, np.linalg.inv(H), v))
    plain = SeparablePriorMetric(L_l, L_s, 10.0, T)
    assert plain.rank == 0 and np.array_equal(plain.root(v), v) and np.array_equal(plain.W(v), v)


@pytest.mark.gpu
def test_separable_sampler_under_its_prior_metric_mixes_and_agrees_with_the_dense_mass_loop():
    """The separable model's sampler call (Separable_model.py:209-210) for B chains, the recipe of tools/hmc_sep.py: mode by
    metric-preconditioned L-BFGS (polish_map_separable), SeparablePriorMetric there, a warm-up, the metric REBUILT at the chains'
    mean (this posterior's mass sits far from its mode along the sigma(x) <-> B scale ridge: the mode's curvature is not the typical
    set's), then the whitened-momentum loop of BatchedHMCSeparable -- against the base class's lock-step loop with the SAME metric
    written out as a dense mass matrix (posterior summaries within Monte-Carlo error), and against the identity mass at the same
    cost (which does not mix)."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMCSeparable, polish_map_separable, separable_prior_metric
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hmc_1000 as H
    N, M = 64, 3
    d = sim.simulate_separable(N, M, 4)
    h = dict(sim.HYPER_SEP)
    p0 = sim.perturb(d["pars_true"], 0.05, 0.4)
    q0, nl, gn, nev = polish_map_separable(d["x"], d["Y"], h, p0, maxiter=400, rounds=8, probes=40, rank=32, batch=8)
    assert gn < 1e-5 * abs(nl) + 1e-3
    P = q0.shape[0]
    met = separable_prior_metric(d["x"], d["Y"], h, q0, rank=32, oversample=8, seed=3, batch=8)
    assert met.info["most_negative"] > -0.9 and met.rank >= 4
    B, L = 8, 10
    init = np.repeat(q0[None], B, 0)
    sw, iw = BatchedHMCSeparable(d["x"], d["Y"], h, init, step_size=0.06, num_steps_in_leap=20, seed=40, M=met, step_jitter=0.2).run(400)
    assert iw["accept_rate"].mean() > 0.6
    cur = sw[-1]
    for wdw in range(2):                                     # two adaptation windows
        met = separable_prior_metric(d["x"], d["Y"], h, sw[-100:].mean((0, 1)), rank=32, oversample=8, seed=5 + wdw, batch=8, factors=met)
        assert met.info["most_negative"] > -0.9
        sw, iw = BatchedHMCSeparable(d["x"], d["Y"], h, cur, step_size=0.15, num_steps_in_leap=L, seed=50 + wdw, M=met, step_jitter=0.2).run(400)
        cur = sw[-1]
    assert iw["accept_rate"].mean() > 0.8
    a = sw[200:]
    rh = H.split_rhat(a)
    assert np.nanmedian(rh) < 1.05 and np.nanquantile(rh, 0.99) < 1.2
    # the same metric as a dense matrix through the base class's loop (momenta p = chol(M) z, velocity M^-1 p), from the same positions
    Lb = met.apply(np.eye(P), False).T
    Wm = np.eye(P) - met.U.T @ np.diag(met.lam / (1 + met.lam)) @ met.U
    Minv = Lb @ Wm @ Lb.T
    Minv = 0.5 * (Minv + Minv.T)
    sdn, idn = BatchedHMCSeparable(d["x"], d["Y"], h, cur, step_size=0.15, num_steps_in_leap=L, seed=77, Minv=Minv).run(400)
    assert abs(idn["accept_rate"].mean() - iw["accept_rate"].mean()) < 0.12
    b = sdn[200:]
    for i in (P - 1, 0, N // 2, N + 5, 2 * N, 2 * N + 1):
        ma, mb = a[:, :, i].mean(), b[:, :, i].mean()
        sd = 0.5 * (a[:, :, i].std() + b[:, :, i].std())
        ess = min(H.multichain_ess(a[:, :, [i]])[0], H.multichain_ess(b[:, :, [i]])[0])
        assert abs(ma - mb) < 5.0 * sd * np.sqrt(2.0 / max(ess, 10.0)), (i, ma, mb, sd, ess)
    # the posterior's mass is far from its mode (what the adaptation windows are for): log-diagonal of L
    assert a[:, :, 2 * N].mean() - q0[2 * N] > 3.0 * a[:, :, 2 * N].std()
    si, ii = BatchedHMCSeparable(d["x"], d["Y"], h, init, step_size=2e-4, num_steps_in_leap=L, seed=40).run(400)
    assert np.nanmedian(H.split_rhat(si[200:])) > 1.5


@pytest.mark.gpu
def test_one_call_recipes_converge_on_both_models():
    """drivers.sample_nonseparable / sample_separable: the whole recipe (mode, metric, warm-up, adaptation windows, step search, main
    run) behind one call, from the start point a script would hand over -- split R-hat of the second half near 1 on every parameter."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hmc_1000 as H
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import sample_nonseparable, sample_separable
    N, M = 64, 3
    log = []
    d = sim.simulate_nonseparable(N, M, seed=5)
    S, info = sample_nonseparable(d["x"], d["Y"], sim.HYPER_SVC, sim.perturb(d["pars_true"], 0.05, 0.7), chains=8, iters=600,
                                  num_steps_in_leap=10, rank=32, windows=1, window_iters=100, step_candidates=(0.08, 0.12, 0.16),
                                  progress=log.append)
    assert S.shape == (600, 8, N * 7 + 1) and info["stages"][-1]["stage"] == "main" and info["stages"][-1]["accept_rate_mean"] > 0.65
    assert info["mode"]["whitened_gradient_norm"] < 1.0 and info["metric"].rank >= 8 and any("step search" in m for m in log)
    rh = H.split_rhat(S[300:])
    assert np.nanmedian(rh) < 1.05 and np.nanquantile(rh, 0.99) < 1.3, (np.nanmedian(rh), np.nanquantile(rh, 0.99), np.nanmax(rh))
    ds = sim.simulate_separable(N, M, 4)
    S2, info2 = sample_separable(ds["x"], ds["Y"], sim.HYPER_SEP, sim.perturb(ds["pars_true"], 0.05, 0.4), chains=8, iters=600,
                                 num_steps_in_leap=10, rank=32, warm=200, windows=2, window_iters=300, batch=8)
    assert S2.shape == (600, 8, 2 * N + 7) and info2["stages"][-1]["accept_rate_mean"] > 0.7
    assert [st["stage"] for st in info2["stages"]][:3] == ["warm-up", "adaptation window 0", "adaptation window 1"]
    rh2 = H.split_rhat(S2[300:])
    assert np.nanmedian(rh2) < 1.06 and np.nanquantile(rh2, 0.99) < 1.3, (np.nanmedian(rh2), np.nanquantile(rh2, 0.99), np.nanmax(rh2))
