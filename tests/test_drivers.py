"""MAP / HMC drivers.  CPU part: the HMC machinery on an analytic Gaussian target;
GPU part: the reference's MAP trajectory fixture and HMC energy conservation on the real potential."""
import os

import numpy as np
import pytest
import torch

from conftest import SVC_KEYS, golden, hyper_dict


def _gauss_potential(A, b):
    At = torch.from_numpy(A)
    bt = torch.from_numpy(b)

    def U(q):
        r = q - bt
        return 0.5 * torch.dot(r, At @ r)
    return U


def test_hmc_on_gaussian_target_cpu():
    from nonstationary_multivariate_gaussian_process_amd.drivers import HMCSampler
    rng = np.random.default_rng(0)
    P = 6
    G = rng.standard_normal((P, P))
    Sigma = G @ G.T / P + 0.3 * np.eye(P)
    A = np.linalg.inv(Sigma)
    b = rng.standard_normal(P)
    h = HMCSampler(sample_size=3000, potential_func=_gauss_potential(A, b), init_position=np.zeros(P), step_size=0.15,
                   num_steps_in_leap=12, seed=1)
    samples, info = h.main_hmc_loop()
    assert samples.shape == (3000, P) and info["accept_rate"] > 0.85
    burn = samples[500:]
    assert np.allclose(burn.mean(0), b, atol=0.15)
    assert np.allclose(np.cov(burn.T), Sigma, atol=0.25)
    # leapfrog is time reversible and nearly energy conserving for a small step
    q0 = rng.standard_normal(P)
    U0, g0 = h.potential_and_grad(q0)
    p0 = rng.standard_normal(P)
    q1, p1, U1, g1 = h.leapfrog(q0, p0, g0, 0.01)
    assert abs((U1 + h.kinetic(p1)) - (U0 + h.kinetic(p0))) < 1e-4
    qb, pb, _, _ = h.leapfrog(q1, -p1, g1, 0.01)
    assert np.allclose(qb, q0, atol=1e-10) and np.allclose(-pb, p0, atol=1e-10)
    # dense mass matrix equal to the precision: one long trajectory still accepts
    h2 = HMCSampler(sample_size=200, potential_func=_gauss_potential(A, b), init_position=b, step_size=0.2,
                    num_steps_in_leap=8, M=A, seed=2, adaptive_step_size=True)
    s2, info2 = h2.main_hmc_loop()
    assert info2["accept_rate"] > 0.7 and s2.shape == (200, P)


def test_lockstep_chains_reject_a_trajectory_that_left_the_domain_cpu():
    """A chain pushed out of the potential's domain MID-trajectory (U = inf at an intermediate leapfrog point, finite again
    at the end point) must be rejected, as the single-chain sampler does; the other chains are unaffected and every
    chain reproduces HMCSampler on the same random stream."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import HMCSampler, LockStepHMC
    P, B, S, eps, L = 3, 4, 30, 0.35, 6
    lo, hi = 0.9, 1.3                              # the potential is undefined in the band lo < |q_0| < hi

    def U_torch(q):
        u = 0.5 * torch.dot(q, q)
        return u + (float("inf") if lo < abs(float(q[0].detach())) < hi else 0.0)

    class Toy(LockStepHMC):
        calls = 0

        def potential_and_grad(self, q):
            U = 0.5 * (q * q).sum(1)
            bad = (np.abs(q[:, 0]) > lo) & (np.abs(q[:, 0]) < hi)
            g = q.copy()
            U[bad] = np.inf
            g[bad] = 0.0                           # what BatchedHMC reports for a failed chain
            return U, g

    init = np.array([[1.45, 0.1, -0.2], [0.0, 0.3, 0.1], [-1.4, 0.0, 0.4], [0.5, -0.5, 0.5]])      # two outside, two inside the band
    lock = Toy(init, step_size=eps, num_steps_in_leap=L, seed=5)
    samples, info = lock.run(S)
    crossed_and_returned = 0
    for b in range(B):
        hs = HMCSampler(sample_size=S, potential_func=U_torch, init_position=init[b], step_size=eps, num_steps_in_leap=L,
                        seed=5 + b)
        single, sinfo = hs.main_hmc_loop()
        assert np.allclose(single, samples[:, b, :], rtol=1e-12, atol=1e-13), b
        assert abs(sinfo["accept_rate"] - info["accept_rate"][b]) < 1e-12
    # the scenario really occurs in this run: replay chain trajectories and count end points that are back inside the domain
    # after an excursion (the old code accepted those)
    lock2 = Toy(init, step_size=eps, num_steps_in_leap=L, seed=5)
    q = lock2.q.copy()
    for it in range(S):
        p = np.stack([r.standard_normal(P) for r in lock2.rngs])
        for r in lock2.rngs:
            r.random()
        U, g = lock2.potential_and_grad(q)
        p1 = p - 0.5 * eps * g
        q1 = q.copy()
        out = np.zeros(B, dtype=bool)
        for step in range(L):
            q1 = q1 + eps * p1
            U1, g1 = lock2.potential_and_grad(q1)
            out |= ~np.isfinite(U1)
            p1 = p1 - (eps if step < L - 1 else 0.5 * eps) * g1
        crossed_and_returned += int(np.sum(out & np.isfinite(U1)))
        q = samples[it]
    assert crossed_and_returned > 0
    a0 = np.abs(samples[:, :, 0])
    assert not np.any((a0 > lo) & (a0 < hi))



def test_lockstep_hmc_with_a_mass_matrix_follows_the_single_chain_sampler_cpu():
    """LockStepHMC with a diagonal and with a dense constant mass matrix (the reference's production sampler passes
    M = inv(sample covariance), Nonseparable_model_mpiKAISER.py:267-270,398-411) against HMCSampler(M=...) chain by chain on a
    Gaussian target, same random streams: momenta chol(M) z, kinetic energy 1/2 p^T M^-1 p, drift q += eps M^-1 p."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import HMCSampler, LockStepHMC
    rng = np.random.default_rng(5)
    P, B, S = 7, 3, 30
    G = rng.standard_normal((P, P))
    A = G @ G.T / P + 0.5 * np.eye(P)
    b = rng.standard_normal(P)
    W = rng.standard_normal((P, P))
    masses = {"diag": np.exp(rng.standard_normal(P)), "dense": W @ W.T / P + np.eye(P)}
    init = rng.standard_normal((B, P))

    class Gauss(LockStepHMC):
        def potential_and_grad(self, q):
            r = q - b
            return 0.5 * np.einsum("bi,ij,bj->b", r, A, r), r @ A

    for kind, Mm in masses.items():
        ls = Gauss(init, step_size=0.2, num_steps_in_leap=6, seed=21, M=Mm)
        samples, info = ls.run(S)
        assert 0.5 < info["accept_rate"].mean() <= 1.0
        for c in range(B):
            hs = HMCSampler(sample_size=S, potential_func=_gauss_potential(A, b), init_position=init[c], step_size=0.2,
                            num_steps_in_leap=6, M=(np.diag(Mm) if kind == "diag" else Mm), seed=0)
            hs.rng = np.random.default_rng(21 + c)
            single, _ = hs.main_hmc_loop()
            assert np.allclose(single, samples[:, c], rtol=1e-10, atol=1e-12), kind


@pytest.mark.gpu
def test_map_trajectory_matches_reference_fixture():
    """100 Adam steps (lr 0.2) from the RNG-free start: the reference's target_value_hist (tests/golden/map_svc_N64_M3,
    generated by running Nonseparable_model.py's loop body on the reference)."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import map_nonseparable
    g = golden("map_svc_N64_M3")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    pars, hist = map_nonseparable(g["x"], g["Y"], g["pars0"], h, N_opt=int(g["steps"]), lr=float(g["lr"]))
    ref = g["target_value_hist"]
    rel = np.abs(hist - ref) / np.abs(ref)
    assert rel[:20].max() < 1e-6, rel[:20]
    assert rel.max() < 1e-4, rel.max()                 # Adam amplifies rounding-level gradient differences over 100 steps
    assert np.linalg.norm(pars - g["pars_end"]) / np.linalg.norm(g["pars_end"]) < 1e-3


@pytest.mark.gpu
def test_hmc_energy_conservation_on_the_nonseparable_potential(tmp_path):
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    from nonstationary_multivariate_gaussian_process_amd.drivers import HMCSampler, map_nonseparable
    g = golden("svc_rngfree_N32_M2")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    x, Y = torch.from_numpy(g["x"]), torch.from_numpy(g["Y"])
    ck = str(tmp_path / "MAP.dat")
    pars, hist = map_nonseparable(g["x"], g["Y"], g["pars"], h, N_opt=30, lr=0.05, checkpoint_path=ck,
                                  checkpoint_every=10)
    import pickle
    assert np.array_equal(pickle.load(open(ck, "rb")), pars)       # MAP.dat layout = the flat float64 vector
    # the call pattern of Nonseparable_model.py:228-231
    hmc = HMCSampler(sample_size=12, potential_func=logpos.nlogpos_obj_SVC, init_position=pars, step_size=1e-4,
                     num_steps_in_leap=20, x=x, Y=Y, duplicate_samples=True, seed=3, **h)
    samples, info = hmc.main_hmc_loop()
    assert samples.shape == (12, pars.shape[0])
    err = np.abs(info["energy_error"])
    assert np.nanmedian(err) < 0.1 and info["accept_rate"] >= 0.5, (err, info["accept_rate"])
    # second-order integrator on the true gradient: halving the step (same trajectory time) quarters the energy error
    q0 = samples[-1]
    U0, g0 = hmc.potential_and_grad(q0)
    p0 = np.random.default_rng(7).standard_normal(q0.shape[0])
    H0 = U0 + hmc.kinetic(p0)
    errs = []
    for eps, L in ((2e-4, 10), (1e-4, 20), (5e-5, 40)):
        hmc.L = L
        q1, p1, U1, _ = hmc.leapfrog(q0, p0, g0, eps)
        errs.append(abs(U1 + hmc.kinetic(p1) - H0))
    assert 2.5 < errs[0] / errs[1] < 6.0 and 2.5 < errs[1] / errs[2] < 6.0, errs


@pytest.mark.gpu
def test_the_scripts_sampler_line_runs_unchanged_on_the_shipped_hmc_sampler():
    """Nonseparable_model.py:24-25,228-231 as written: `import HMC_Sampler` (the authors' external package, absent from the
    reference tree) resolves to the mirror's HMC_Sampler package once the alias is installed, and the script's sampler call --
    same keywords, incl. TensorType -- returns (samples [sample_size, P], _)."""
    import sys
    import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
    saved = {k: sys.modules.get(k) for k in ("HMC_Sampler", "HMC_Sampler.HMC_sampler")}
    try:
        nmgp_amd.install_utility_alias()
        nmgp_amd.install_hmc_sampler(force=True)
        import HMC_Sampler
        from Utility import logpos
        from Utility import settings
        g = golden("svc_rngfree_N32_M2")
        hyper_pars = hyper_dict(g["hyper"], SVC_KEYS)
        x, Y = torch.from_numpy(g["x"]).type(settings.torchType), torch.from_numpy(g["Y"]).type(settings.torchType)
        estPars, N_hmc = g["pars"], 5
        hmc = HMC_Sampler.HMC_sampler.sampler(sample_size=N_hmc, potential_func=logpos.nlogpos_obj_SVC, init_position=estPars,
                                              step_size=1e-4, num_steps_in_leap=20, x=x, Y=Y, duplicate_samples=True, TensorType=settings.torchType,
                                              **hyper_pars)
        sample, _ = hmc.main_hmc_loop()
        assert sample.shape == (N_hmc, estPars.shape[0]) and np.all(np.isfinite(sample))
        assert np.linalg.norm(sample[-1] - estPars) > 0          # it moved
        # the production call of the MPI variants (Nonseparable_model_mpiKAISER.py:267-270): mass matrix + adaptive step size
        P = estPars.shape[0]
        hmc2 = HMC_Sampler.HMC_sampler.sampler(sample_size=3, potential_func=logpos.nlogpos_obj_SVC, init_position=torch.from_numpy(estPars),
                                               step_size=1e-4, adaptive_step_size=True, num_steps_in_leap=5, M=np.eye(P), x=x, Y=Y,
                                               duplicate_samples=True, TensorType=settings.torchType, **hyper_pars)
        s2, info2 = hmc2.main_hmc_loop()
        assert s2.shape == (3, P) and info2["step_size"] > 0
        # opt-in for the UNCHANGED script: NMGP_HMC_RECIPE=1 turns the same call into the recipe under which the chains converge
        # (mode from init_position, prior-factor metric, warm-up, step search, NMGP_HMC_CHAINS chains on the GPU); same return shape
        os.environ["NMGP_HMC_RECIPE"], os.environ["NMGP_HMC_CHAINS"] = "1", "4"
        try:
            hmc3 = HMC_Sampler.HMC_sampler.sampler(sample_size=40, potential_func=logpos.nlogpos_obj_SVC, init_position=estPars,
                                                   step_size=1e-4, num_steps_in_leap=10, x=x, Y=Y, duplicate_samples=True,
                                                   TensorType=settings.torchType, **hyper_pars)
            s3, info3 = hmc3.main_hmc_loop()
        finally:
            del os.environ["NMGP_HMC_RECIPE"], os.environ["NMGP_HMC_CHAINS"]
        assert s3.shape == (40, P) and info3["all_chains"].shape == (40, 4, P) and np.array_equal(s3, info3["all_chains"][:, 0])
        assert info3["accept_rate"] > 0.5 and info3["step_size"] >= 0.05 and info3["recipe"]["metric"].rank >= 4
        assert np.sqrt(np.mean((s3[-1] - estPars) ** 2)) > 20 * np.sqrt(np.mean((sample[-1] - estPars) ** 2))      # it goes places
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.gpu
def test_batched_hmc_chains_follow_the_single_chain_sampler():
    """Lock-step batched chains (one launch sequence per leapfrog step for all chains) reproduce the trajectories of
    the single-chain sampler started from the same state with the same random stream."""
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, HMCSampler
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_rngfree_N32_M2")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    B, S = 3, 4
    init = np.stack([sim.perturb(g["pars"], 0.01 * b, 0.5 * b) for b in range(B)])
    bh = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-5, num_steps_in_leap=8, seed=11)
    samples, info = bh.run(S)
    assert samples.shape == (S, B, init.shape[1]) and np.all(info["accept_rate"] >= 0.5)
    assert np.nanmax(np.abs(info["energy_error"])) < 1.0
    x, Y = torch.from_numpy(g["x"]), torch.from_numpy(g["Y"])
    for b in range(B):
        rng = np.random.default_rng(11 + b)
        hs = HMCSampler(sample_size=S, potential_func=logpos.nlogpos_obj_SVC, init_position=init[b], step_size=5e-5,
                        num_steps_in_leap=8, x=x, Y=Y, seed=0, **h)
        # same random stream order as BatchedHMC: momentum draw, then the uniform of the accept test
        hs.rng = rng
        single, _ = hs.main_hmc_loop()
        assert np.allclose(single, samples[:, b, :], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_device_resident_trajectories_equal_the_host_lock_step_loop_bit_for_bit():
    """BatchedHMC keeps positions, momenta and gradients in HBM for a whole trajectory (nmgp_svc_batch_traj: elementwise
    leapfrog kernels between the batched evaluations, per-chain validity flags on the device, rejected chains restored by
    nmgp_svc_batch_traj_commit).  Same arithmetic, same random streams as the host-side loop: samples, acceptance and
    energy errors must be IDENTICAL -- for ordinary chains, for a chain whose start point is undefined (it must never move)
    and across rejections (a step size large enough that some proposals are refused)."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_rngfree_N32_M2")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    B, S = 5, 6
    init = np.stack([sim.perturb(g["pars"], 0.01 * b, 0.5 * b) for b in range(B)])
    init[3, 0] = np.nan                       # chain 3: potential undefined from the start
    rates = []
    for eps, L in ((5e-5, 8), (2e-4, 6), (5e-4, 5), (2e-3, 5)):
        runs = []
        for dev in (True, False):
            bh = BatchedHMC(g["x"], g["Y"], h, init, step_size=eps, num_steps_in_leap=L, seed=3, device_resident=dev)
            runs.append(bh.run(S))
        (sd, idv), (sh, ih) = runs
        assert np.array_equal(sd, sh, equal_nan=True)
        assert np.array_equal(idv["accept_rate"], ih["accept_rate"])
        assert np.array_equal(idv["energy_error"], ih["energy_error"], equal_nan=True)
        assert idv["accept_rate"][3] == 0.0 and np.array_equal(sd[-1, 3], init[3], equal_nan=True)
        rates.append(idv["accept_rate"][[0, 1, 2, 4]])
    rates = np.concatenate(rates)
    assert rates.max() == 1.0 and rates.min() < 1.0, rates          # accepted and rejected proposals both occurred
    # A chain that leaves the domain in the MIDDLE of a trajectory (finite potential at the first leapfrog points, undefined
    # later): the device flags it at that step, skips its momentum kicks from there on and the proposal is rejected -- the same
    # decisions and the same bits as the host loop, whose evaluations are logged here to prove where the failure happened.
    class Logged(BatchedHMC):
        def potential_and_grad(self, q):
            U, gr = super().potential_and_grad(q)
            self.log.append(np.isfinite(U).copy())
            return U, gr
    eps, L = 5e-3, 6
    init2 = np.stack([sim.perturb(g["pars"], 0.01 * b, 0.5 * b) for b in range(B)])
    host = Logged(g["x"], g["Y"], h, init2, step_size=eps, num_steps_in_leap=L, seed=3, device_resident=False)
    host.log = []
    sh, ih = host.run(S)
    log = np.array(host.log[1:]).reshape(S, L, B)
    first_bad = np.where(log.all(1), L, np.argmin(log, axis=1))       # [S, B]: first undefined leapfrog point of the trajectory
    assert np.any((first_bad >= 1) & (first_bad < L)), first_bad      # finite at step 0, undefined at an intermediate step
    dev = BatchedHMC(g["x"], g["Y"], h, init2, step_size=eps, num_steps_in_leap=L, seed=3)
    sd, idv = dev.run(S)
    assert np.array_equal(sd, sh) and np.array_equal(idv["accept_rate"], ih["accept_rate"])
    assert np.array_equal(idv["energy_error"], ih["energy_error"], equal_nan=True)
    assert np.all(idv["accept_rate"][first_bad.min(0) < L] < 1.0)     # such proposals are never accepted


@pytest.mark.gpu
def test_batched_hmc_with_a_mass_matrix_on_the_device():
    """nmgp_svc_batch_traj_set_mass: the drift of the device-resident trajectories becomes q += eps M^-1 p -- elementwise for a
    diagonal mass matrix (bit-identical to the host lock-step loop), one GEMM per leapfrog step for a dense one (rounding-level
    agreement) -- and every chain follows HMCSampler(M=...) started from the same state with the same random stream."""
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, HMCSampler
    from nonstationary_multivariate_gaussian_process_amd import sim
    g = golden("svc_rngfree_N32_M2")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    B, S, L = 3, 4, 6
    init = np.stack([sim.perturb(g["pars"], 0.01 * b, 0.5 * b) for b in range(B)])
    P = init.shape[1]
    rng = np.random.default_rng(8)
    W = rng.standard_normal((P, 4))
    dense = 1e4 * (np.eye(P) + 0.3 * W @ W.T / 4)                  # M ~ 1e4: velocities 1e-2 p, so the step can be 100 x larger
    diag = 1e4 * np.exp(0.5 * rng.standard_normal(P))
    x, Y = torch.from_numpy(g["x"]), torch.from_numpy(g["Y"])
    for kind, Mm in (("diag", diag), ("dense", dense)):
        dev = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-3, num_steps_in_leap=L, seed=11, M=Mm, device_momenta=False)
        sd, idv = dev.run(S)
        host = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-3, num_steps_in_leap=L, seed=11, M=Mm, device_resident=False)
        sh, ih = host.run(S)
        # the default with a mass matrix: momenta p0 = chol(M) z and the end point's kinetic energy formed on the device
        # (nmgp_svc_batch_traj_z) -- same random streams, same trajectories; only the summation order of the energies differs
        dz = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-3, num_steps_in_leap=L, seed=11, M=Mm)
        assert dz.device_momenta
        sz, iz = dz.run(S)
        assert np.allclose(sz, sh, rtol=1e-10, atol=1e-12), kind
        assert np.allclose(iz["energy_error"], ih["energy_error"], rtol=0, atol=1e-8) and np.array_equal(iz["accept_rate"], ih["accept_rate"])
        assert np.all(idv["accept_rate"] >= 0.5) and np.nanmax(np.abs(idv["energy_error"])) < 1.0
        assert not np.allclose(sd[-1], init)                        # the chains moved
        if kind == "diag":
            assert np.array_equal(sd, sh) and np.array_equal(idv["energy_error"], ih["energy_error"])
        else:
            assert np.allclose(sd, sh, rtol=1e-10, atol=1e-12)
        for b in range(B):
            hs = HMCSampler(sample_size=S, potential_func=logpos.nlogpos_obj_SVC, init_position=init[b], step_size=5e-3,
                            num_steps_in_leap=L, M=(np.diag(Mm) if kind == "diag" else Mm), x=x, Y=Y, seed=0, **h)
            hs.rng = np.random.default_rng(11 + b)
            single, _ = hs.main_hmc_loop()
            assert np.allclose(single, sd[:, b, :], rtol=1e-9, atol=1e-11), (kind, b)
    # back to the identity: the mass matrix is per-run state of the context
    ident = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-5, num_steps_in_leap=L, seed=11)
    assert not ident.device_momenta
    si, _ = ident.run(2)
    ref = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-5, num_steps_in_leap=L, seed=11, device_resident=False)
    sr, _ = ref.run(2)
    assert np.array_equal(si, sr)
    # ... and the identity with device-side energies: p0 = z
    iz = BatchedHMC(g["x"], g["Y"], h, init, step_size=5e-5, num_steps_in_leap=L, seed=11, device_momenta=True)
    sz, _ = iz.run(2)
    assert np.allclose(sz, sr, rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_trajectory_state_is_invalidated_by_a_new_metric_or_new_subjects():
    """The resident trajectory state (positions, gradients, flags) belongs to the metric and the data it was begun under:
    nmgp_svc_batch_traj_set_mass after nmgp_svc_batch_traj_begin, or new subjects after the start evaluation, must make the next
    trajectory call fail with a state error instead of continuing with a stale gradient; momenta cannot be drawn on the device
    before chol(M) is there."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    g = golden("svc_rngfree_N32_M2")
    hv = g["hyper"]
    B = 2
    c = _lib.Context(0)
    try:
        c.set_data(g["x"], g["Y"])
        c.svc_batch_alloc(B)
        q = np.stack([sim.perturb(g["pars"], 0.01, 0.3 * b) for b in range(B)])
        P = q.shape[1]
        z = np.random.default_rng(0).standard_normal((B, P))

        def start():
            c.svc_batch_set_pars(q)
            c.svc_batch_eval(hv, True, want_grad=True)
            c.svc_batch_traj_begin()
        start()
        q1, p1, U1, failed = c.svc_batch_traj(hv, True, 1e-5, 2, z)          # fine
        assert np.all(np.isfinite(U1)) and not failed.any()
        # a new metric after begin: the trajectory must be begun again
        c.svc_batch_traj_set_mass(np.full(P, 2.0))
        with pytest.raises(_lib.NmgpError, match="traj_begin"):
            c.svc_batch_traj(hv, True, 1e-5, 2, z)
        start()
        # device-side momenta need chol(M) for the metric in force
        with pytest.raises(_lib.NmgpError, match="chol"):
            c.svc_batch_traj_z(hv, True, 1e-5, 2, z)
        c.svc_batch_traj_set_mass_chol(np.full(P, np.sqrt(0.5)))
        qz, kin, Uz, fz = c.svc_batch_traj_z(hv, True, 1e-5, 2, z)
        assert np.all(np.isfinite(kin)) and np.all(kin > 0) and not fz.any()
        c.svc_batch_traj_commit(np.ones(B, dtype=bool))
        # new subjects: the resident gradient no longer belongs to the batch -- begin needs a fresh evaluation
        c.svc_batch_set_subjects(np.stack([g["x"]] * B), np.stack([g["Y"]] * B))
        with pytest.raises(_lib.NmgpError, match="value\\+gradient evaluation"):
            c.svc_batch_traj_begin()
        c.svc_batch_traj_set_mass(None)
        start()
        q2, _, U2, f2 = c.svc_batch_traj(hv, True, 1e-5, 2, z)
        assert np.all(np.isfinite(U2)) and not f2.any()
    finally:
        c.close()


@pytest.mark.gpu
def test_allocating_the_same_batch_size_again_keeps_the_buffers_and_resets_the_state():
    """Every sampler / optimiser object starts with nmgp_svc_batch_alloc(B); with an unchanged B the call must not free and
    re-allocate the batch (116 GB of gradient workspace at the headline size: 5 s of every sampler construction in round 4), but it
    must hand back a FRESH batch: zero parameters, identity metric, no trajectory begun, chains of the resident subject."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    g = golden("svc_rngfree_N32_M2")
    hv = g["hyper"]
    B = 3
    c = _lib.Context(0)
    try:
        c.set_data(g["x"], g["Y"])
        c.svc_batch_alloc(B)
        q = np.stack([sim.perturb(g["pars"], 0.01, 0.3 * b) for b in range(B)])
        P = q.shape[1]

        def evaluate():
            c.svc_batch_set_pars(q)
            c.svc_batch_eval(hv, True, want_grad=True)
            out, st = c.svc_batch_fetch()
            return out, c.svc_batch_fetch_grad()
        out1, g1 = evaluate()
        dev1 = c.lib.nmgp_svc_batch_pars_dev(c.h)
        c.svc_batch_traj_set_mass(np.full(P, 2.0))
        c.svc_batch_set_subjects(np.stack([g["x"]] * B), np.stack([g["Y"]] * B))
        c.svc_batch_alloc(B)                                  # the same size again
        assert c.lib.nmgp_svc_batch_pars_dev(c.h) == dev1     # same buffers ...
        assert np.all(c.svc_batch_get_pars() == 0.0)          # ... fresh state
        with pytest.raises(_lib.NmgpError, match="value\\+gradient evaluation"):
            c.svc_batch_traj_begin()
        out2, g2 = evaluate()                                 # chains of the resident subject again, bit for bit
        assert np.array_equal(out1, out2) and np.array_equal(g1, g2)
        c.svc_batch_traj_begin()
        z = np.random.default_rng(0).standard_normal((B, P))
        q1, kin, U1, failed = c.svc_batch_traj_z(hv, True, 1e-5, 2, z)      # identity metric: no chol(M) needed
        assert not failed.any() and np.allclose(kin, 0.5 * (z * z).sum(1), rtol=1e-3)
        c.svc_batch_alloc(B + 1)                              # another size: a new batch
        assert np.all(c.svc_batch_get_pars() == 0.0) and c.svc_batch_get_pars().shape == (B + 1, P)
    finally:
        c.close()


@pytest.mark.gpu
def test_batched_separable_hmc_chains_follow_the_single_chain_sampler():
    """BatchedHMCSeparable: B chains of the separable model in lock-step on nmgp_sep_batch_eval; every chain reproduces
    HMCSampler(potential_func=logpos.nlogpos_obj, ...) -- the sampler call of Separable_model.py:209 -- from the same state with the
    same random stream."""
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMCSeparable, HMCSampler
    from nonstationary_multivariate_gaussian_process_amd import sim
    from conftest import SEP_KEYS
    g = golden("sep_rngfree_N32_M2")
    h = hyper_dict(g["hyper"], SEP_KEYS)
    B, S, L = 3, 4, 5
    init = np.stack([sim.perturb(g["pars"], 0.01 * b, 0.5 * b) for b in range(B)])
    bh = BatchedHMCSeparable(g["x"], g["Y"], h, init, step_size=2e-4, num_steps_in_leap=L, seed=21)
    sb, ib = bh.run(S)
    assert sb.shape == (S, B, init.shape[1]) and np.all(ib["accept_rate"] >= 0.5) and not np.allclose(sb[-1], init)
    x, Y = torch.from_numpy(g["x"]), torch.from_numpy(g["Y"])
    for b in range(B):
        hs = HMCSampler(sample_size=S, potential_func=logpos.nlogpos_obj, init_position=init[b], step_size=2e-4, num_steps_in_leap=L,
                        Y=Y, x=x, seed=0, **h)
        hs.rng = np.random.default_rng(21 + b)
        single, _ = hs.main_hmc_loop()
        assert np.allclose(single, sb[:, b, :], rtol=1e-9, atol=1e-11), b


@pytest.mark.gpu
def test_one_hmc_chain_per_subject_equals_the_subjects_sampled_one_at_a_time():
    """BatchedHMC over a multi-subject batch (x [B, N], Y [B, N, M]: config 4's unit, every subject with its own data and
    prior factors) against the same sampler run on each subject alone with the same random stream."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC
    from nonstationary_multivariate_gaussian_process_amd import sim
    B, S, N, M = 3, 4, 48, 3
    subs = [sim.simulate_nonseparable(N, M, seed=20 + b) for b in range(B)]
    xs, Ys = np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs])
    init = np.stack([sim.perturb(d["pars_true"], 0.02, 0.3 * b) for b, d in enumerate(subs)])
    h = sim.HYPER_SVC_MPISIM
    allb = BatchedHMC(xs, Ys, h, init, step_size=5e-5, num_steps_in_leap=6, seed=5)
    samples, info = allb.run(S)
    assert np.all(info["accept_rate"] > 0)
    for b in range(B):
        one = BatchedHMC(subs[b]["x"], subs[b]["Y"], h, init[b:b + 1], step_size=5e-5, num_steps_in_leap=6, seed=5 + b)
        sb, _ = one.run(S)
        assert np.allclose(sb[:, 0], samples[:, b], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_several_hmc_chains_per_subject_share_the_subject_on_the_device():
    """BatchedHMC(xs [S, N], Ys, chains_per_subject=k): S subjects x k chains in ONE batch, the chains of a subject sharing its
    inputs and GP-prior factors (nmgp_svc_batch_set_subjects_chains).  Chain (s, k) must equal that chain run alone on its subject
    with the same random stream; value-only and value+gradient evaluations of the mixed batch equal single evaluations."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC
    S_, K, N, M, samples = 3, 2, 48, 3, 3
    subs = [sim.simulate_nonseparable(N, M, seed=30 + s) for s in range(S_)]
    xs, Ys = np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs])
    init = np.stack([sim.perturb(subs[s]["pars_true"], 0.02, 0.3 * s + 0.7 * k) for s in range(S_) for k in range(K)])
    h = sim.HYPER_SVC_MPISIM
    hv = [h[k] for k in SVC_KEYS]
    # one evaluation of the mixed batch against single-subject evaluations
    c = _lib.Context(0)
    try:
        c.set_data(xs[0], Ys[0])
        c.svc_batch_alloc(S_ * K)
        c.svc_batch_set_subjects(xs, Ys, K)
        c.svc_batch_set_pars(init)
        c.svc_batch_eval(hv, True, True)
        out, st = c.svc_batch_fetch()
        grads = c.svc_batch_fetch_grad()
        assert np.all(st == 0)
        for s in range(S_):
            c.set_data(xs[s], Ys[s])
            for k in range(K):
                o1, g1 = c.logpos_svc(init[s * K + k], hv, prior=True, want_grad=True)
                # (the batch solves the prior systems by substitution, the single evaluation through the library: kappa * eps apart)
                assert np.allclose(out[s * K + k], o1, rtol=1e-10, atol=0)
                assert np.linalg.norm(grads[s * K + k] - g1) < 1e-7 * np.linalg.norm(g1)
    finally:
        c.close()
    allb = BatchedHMC(xs, Ys, h, init, step_size=5e-5, num_steps_in_leap=6, seed=5, chains_per_subject=K)
    smp, info = allb.run(samples)
    assert np.all(info["accept_rate"] > 0)
    for s in range(S_):
        for k in range(K):
            b = s * K + k
            one = BatchedHMC(subs[s]["x"], subs[s]["Y"], h, init[b:b + 1], step_size=5e-5, num_steps_in_leap=6, seed=5 + b)
            sb, _ = one.run(samples)
            assert np.allclose(sb[:, 0], smp[:, b], rtol=1e-9, atol=1e-11), (s, k)


@pytest.mark.gpu
def test_map_then_hmc_with_the_sample_covariance_as_mass_matrix_then_prediction():
    """The reference's production workflow in one piece, small (Nonseparable_model_mpiKAISER.py:356, 398-411, 267-270; prediction
    Nonseparable_model.py:333): MAP by Adam -> a first HMC run from the MAP point (identity mass) -> its sample covariance becomes
    the mass matrix, M = inv(sample_cov + 1e-10 I), step size raised, 5 leapfrog steps -> prediction on a grid at the posterior mean.
    Checks that the pieces fit (shapes, finiteness, the device-resident dense-mass path = the host loop) and that the preconditioned
    sampler moves several times as far per sample as the first one (step 0.25 against 3e-4)."""
    from nonstationary_multivariate_gaussian_process_amd import _lib, sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedHMC, BatchedMAP
    N, M, B = 24, 2, 8
    d = sim.simulate_nonseparable(N, M, seed=41)
    h = sim.HYPER_SVC
    p0 = sim.perturb(d["pars_true"], 0.05, 0.3)
    pars, hist, alive = BatchedMAP(d["x"][None], d["Y"][None], h, p0[None], lr=0.05).run(300)
    assert alive[0] and hist[-1, 0] > hist[0, 0]                     # the log posterior went up
    q0 = np.repeat(pars, B, axis=0)
    first = BatchedHMC(d["x"], d["Y"], h, q0, step_size=3e-4, num_steps_in_leap=10, seed=7)     # (1e-3 is already rejected: stiff priors)
    s1, i1 = first.run(40)
    assert np.mean(i1["accept_rate"]) > 0.3
    flat = s1[10:].reshape(-1, s1.shape[-1])                          # [samples x chains, P]
    P = flat.shape[1]
    assert flat.shape[0] > P
    cov = np.cov(flat.T) + 1e-10 * np.eye(P)                          # mpiKAISER.py:405
    Mh = np.linalg.inv(cov)                                           # mpiKAISER.py:406: M_hmc = inv(sample_cov)
    runs = []
    for dev in (True, False):
        hm = BatchedHMC(d["x"], d["Y"], h, s1[-1], step_size=0.25, num_steps_in_leap=5, seed=9, M=Mh, Minv=cov, device_resident=dev)
        runs.append(hm.run(20))
    (s2, i2), (s2h, i2h) = runs
    assert np.allclose(s2, s2h, rtol=1e-8, atol=1e-10)                # dense mass matrix on the device = the host loop
    assert np.all(np.isfinite(s2)) and np.mean(i2["accept_rate"]) > 0.3
    move1 = np.sqrt(np.mean((s1[-1] - s1[-2]) ** 2)) + 1e-300
    move2 = np.sqrt(np.mean((s2[-1] - s2[-2]) ** 2))
    assert move2 > 2.0 * move1       # measured: 0.0088 against 0.0020 per sample, with half the gradient evaluations per sample
    # prediction at the posterior mean on the reference's grid
    post = s2.reshape(-1, P).mean(0)
    c = _lib.Context(0)
    try:
        c.set_data(d["x"], d["Y"])
        mean, var, Ls = c.predict_svc(post, [h[k] for k in SVC_KEYS], np.linspace(0.0, 1.0, 201))
    finally:
        c.close()
    assert mean.shape == (201, M) and np.all(np.isfinite(mean)) and np.all(var > 0) and Ls.shape == (201, M * (M + 1) // 2)


@pytest.mark.gpu
def test_config1_stationary_200_mcmc_iterations_follow_the_cpu_oracle():
    """BASELINE config 1 (stationary GP, D = 2, N = 128, 200 MCMC iterations -- the reference's CPU-runnable case): the same
    HMC loop, random stream and start, once with the MI355X potential (``logpos.nlogpos_obj_S`` through the mirror) and once with
    the CPU oracle's value + gradient.  Both chains must agree sample for sample while rounding-level gradient differences stay
    small (the first 50 iterations, 1e-6 relative), keep the same accept decisions, and end with the same posterior means."""
    import torch
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    from nonstationary_multivariate_gaussian_process_amd.drivers import HMCSampler
    from oracle import nmgp_oracle as O
    g = golden("sta_sim_N128_M2")
    h = sim.HYPER_STA
    x, Y = torch.from_numpy(g["x"]), torch.from_numpy(g["Y"])
    S, L, eps = 200, 5, 2e-2

    class OracleSampler(HMCSampler):
        def potential_and_grad(self, q):
            r, grad = O.nlogpos_obj_S(q, g["Y"], g["x"], **h, grad=True)
            return float(r), np.asarray(grad, dtype=np.float64).copy()

    gpu = HMCSampler(sample_size=S, potential_func=logpos.nlogpos_obj_S, init_position=g["pars"], step_size=eps,
                     num_steps_in_leap=L, x=x, Y=Y, seed=9, **h)
    cpu = OracleSampler(sample_size=S, potential_func=None, init_position=g["pars"], step_size=eps, num_steps_in_leap=L, seed=9)
    sg, ig = gpu.main_hmc_loop()
    sc, ic = cpu.main_hmc_loop()
    assert sg.shape == sc.shape == (S, g["pars"].shape[0])
    assert 0.3 < ig["accept_rate"] <= 1.0
    rel = np.abs(sg - sc).max(1) / np.abs(sc).max(1)
    assert rel[:50].max() < 1e-6, rel[:50].max()
    assert abs(ig["accept_rate"] - ic["accept_rate"]) <= 2.0 / S
    assert np.allclose(sg[100:].mean(0), sc[100:].mean(0), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_separable_and_stationary_map_trajectories_match_the_reference_fixture():
    """The callers of configs 5 and 1: the Adam loops of Separable_model.py:147-166 and Stationary_model.py:112-131, 60 steps
    from the RNG-free start; the reference's own target_value_hist (tests/golden/map_sep_sta_N64_M3.npz)."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import map_separable, map_stationary
    from conftest import SEP_KEYS, STA_KEYS
    g = golden("map_sep_sta_N64_M3")
    steps = int(g["steps"])
    pars, hist = map_separable(g["x"], g["Y"], g["sep_pars0"], hyper_dict(g["sep_hyper"], SEP_KEYS), N_opt=steps)
    rel = np.abs(hist - g["sep_hist"]) / np.abs(g["sep_hist"])
    # (the CPU oracle's analytic gradient follows the reference's trajectory to 4e-7 over these 60 steps)
    assert rel.max() < 1e-5, rel.max()
    assert np.linalg.norm(pars - g["sep_end"]) / np.linalg.norm(g["sep_end"]) < 1e-3
    pars, hist = map_stationary(g["x"], g["Y"], g["sta_pars0"], hyper_dict(g["sta_hyper"], STA_KEYS), N_opt=steps)
    rel = np.abs(hist - g["sta_hist"]) / np.abs(g["sta_hist"])
    assert rel.max() < 1e-7, rel.max()
    assert pars[1] == g["sta_pars0"][1]                      # tilde_sigma is not optimised
    assert np.linalg.norm(pars - g["sta_end"]) / np.linalg.norm(g["sta_end"]) < 1e-3


def test_lockstep_map_reproduces_torch_adam_and_isolates_a_failing_subject_cpu():
    """LockStepMAP's update rule against torch.optim.Adam on a quadratic-plus-quartic toy objective (bit-level agreement is not
    promised, 1e-12 is), and a subject whose evaluation starts failing is frozen while the others go on."""
    from nonstationary_multivariate_gaussian_process_amd.drivers import LockStepMAP
    rng = np.random.default_rng(3)
    B, P, steps = 3, 7, 40
    A = rng.standard_normal((B, P, P))
    A = np.einsum("bij,bkj->bik", A, A) / P + np.eye(P)
    p0 = rng.standard_normal((B, P))

    def f_torch(p, a):
        return 0.5 * p @ (a @ p) + 0.1 * (p ** 4).sum()

    class Toy(LockStepMAP):
        def value_and_grad(self, Pm):
            out = np.zeros((B, 5))
            g = np.zeros((B, P))
            st = np.zeros(B, dtype=np.int32)
            for b in range(B):
                out[b, 0] = 0.5 * Pm[b] @ (A[b] @ Pm[b]) + 0.1 * (Pm[b] ** 4).sum()
                g[b] = A[b] @ Pm[b] + 0.4 * Pm[b] ** 3
            if self.t >= 10:
                st[1] = 5                      # subject 1 breaks from iteration 10 on
            return out, g, st

    lm = Toy(p0, lr=0.2)
    pars, hist, alive = lm.run(steps)
    assert list(alive) == [True, False, True] and np.all(np.isinf(hist[10:, 1])) and np.all(np.isfinite(hist[:10, 1]))
    for b in (0, 2):
        p = torch.from_numpy(p0[b].copy()).requires_grad_(True)
        opt = torch.optim.Adam([p], lr=0.2)
        a = torch.from_numpy(A[b])
        ref = []
        for _ in range(steps):
            opt.zero_grad()
            v = f_torch(p, a)
            v.backward()
            opt.step()
            ref.append(-float(v.detach()))
        assert np.allclose(hist[:, b], ref, rtol=1e-12, atol=1e-12)
        assert np.allclose(pars[b], p.detach().numpy(), rtol=1e-11, atol=1e-12)


@pytest.mark.gpu
def test_batched_map_of_several_subjects_matches_the_reference_trajectory_and_single_subject_runs():
    """Config 4's caller loop: three subjects advance together, one batched value+gradient launch sequence per Adam iteration.
    Subject 0 is the reference's MAP fixture (its own target_value_hist); the others must follow single-subject runs."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedMAP, map_nonseparable
    g = golden("map_svc_N64_M3")
    h = hyper_dict(g["hyper"], SVC_KEYS)
    N, M = g["Y"].shape
    steps = 40
    subs = [sim.simulate_nonseparable(N, M, seed=s) for s in (11, 12)]
    xs = np.stack([g["x"]] + [d["x"] for d in subs])
    Ys = np.stack([g["Y"]] + [d["Y"] for d in subs])
    p0 = np.stack([g["pars0"]] + [sim.perturb(d["pars_true"], 0.05, 0.3) for d in subs])
    bm = BatchedMAP(xs, Ys, h, p0, lr=float(g["lr"]))
    pars, hist, alive = bm.run(steps)
    assert np.all(alive)
    ref = g["target_value_hist"][:steps]
    rel = np.abs(hist[:, 0] - ref) / np.abs(ref)
    assert rel[:20].max() < 1e-6 and rel.max() < 1e-4, rel.max()
    for k in (1, 2):
        ps, hs = map_nonseparable(xs[k], Ys[k], p0[k], h, N_opt=steps, lr=float(g["lr"]))
        r = np.abs(hist[:, k] - hs) / np.abs(hs)
        assert r.max() < 1e-7, (k, r.max())
        assert np.linalg.norm(pars[k] - ps) / np.linalg.norm(ps) < 1e-6


@pytest.mark.gpu
def test_device_resident_adam_equals_the_host_update_bit_for_bit():
    """BatchedMAP keeps parameters, gradients and Adam's moments in HBM (nmgp_svc_batch_adam_step).  Same arithmetic as the
    host-side row-by-row update: trajectories (target_value_hist), final parameters and the alive flags must be IDENTICAL, also
    when one subject's evaluation fails along the way (it is frozen, the others go on)."""
    from nonstationary_multivariate_gaussian_process_amd import sim
    from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedMAP
    N, M, B, steps = 40, 3, 4, 25
    subs = [sim.simulate_nonseparable(N, M, seed=30 + b) for b in range(B)]
    xs, Ys = np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs])
    p0 = np.stack([sim.perturb(d["pars_true"], 0.05, 0.2 + 0.1 * b) for b, d in enumerate(subs)])
    p0[2, 5] = np.nan                          # subject 2: undefined from the first evaluation on
    res = []
    for dev in (True, False):
        bm = BatchedMAP(xs, Ys, sim.HYPER_SVC_MPISIM, p0, lr=1e-2, device_resident=dev)
        res.append(bm.run(steps))
    (pd_, hd, ad), (ph, hh, ah) = res
    assert np.array_equal(ad, ah) and list(ad) == [True, True, False, True]
    assert np.array_equal(hd, hh) and np.all(np.isneginf(hd[:, 2]))
    assert np.array_equal(pd_, ph, equal_nan=True)
    assert np.array_equal(pd_[2], p0[2], equal_nan=True)          # frozen where it failed
    assert np.all(hd[-1, [0, 1, 3]] > hd[0, [0, 1, 3]])            # and the others improved their posterior
