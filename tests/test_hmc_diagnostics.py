"""The convergence diagnostics tools/hmc_1000.py reports for BASELINE config 3 (CPU): the multi-chain effective sample size must NOT be
a sum of per-chain figures -- chains that sit at different places have a small ESS however smooth each of them looks."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_multichain_ess_and_split_rhat_on_known_processes():
    import hmc_1000 as H
    rng = np.random.default_rng(0)
    S, C, K = 600, 8, 4
    iid = rng.standard_normal((S, C, K))
    ess = H.multichain_ess(iid)
    assert np.all(ess > 0.7 * S * C) and np.all(ess < 1.3 * S * C)
    assert np.all(np.abs(H.split_rhat(iid) - 1.0) < 0.01)
    # AR(1) with rho = 0.9: ESS = n (1 - rho) / (1 + rho)
    ar = np.zeros((S, C, K))
    e = rng.standard_normal((S, C, K)) * np.sqrt(1 - 0.81)
    for t in range(1, S):
        ar[t] = 0.9 * ar[t - 1] + e[t]
    ess = H.multichain_ess(ar)
    want = S * C * 0.1 / 1.9
    assert np.all(ess > 0.5 * want) and np.all(ess < 2.0 * want)
    # chains that have not mixed: every chain is white noise around its OWN level -- the per-chain sum (round 4's figure) says
    # "thousands", the multi-chain estimator says "a handful", split-R-hat is far from 1
    stuck = iid + 2.0 * np.arange(C)[None, :, None]
    assert np.all(H.multichain_ess(stuck) < 3 * C) and np.all(H.autocorr_ess(stuck) > 0.5 * S * C)
    assert np.all(H.split_rhat(stuck) > 2.0)
    # a slow common drift in every chain is seen by the split halves
    drift = iid + np.linspace(0, 3, S)[:, None, None]
    assert np.all(H.split_rhat(drift) > 1.2)


def test_randomized_eigs_and_preconditioned_lbfgs_on_synthetic_operators_cpu():
    """The two host-side numerical pieces behind drivers.prior_lowrank_metric / polish_map, on problems with known answers:
    (1) leading eigenpairs BY MAGNITUDE of a symmetric operator given only through products with blocks of vectors -- a negative
    eigenvalue beyond the threshold enters as |lam| (the SoftAbs rule); (2) L-BFGS whose initial matrix is (I + U lam U^T)^-1: on
    f(w) = 1/2 w^T (I + U lam U^T) w - b^T w + a mild quartic it converges in fewer evaluations than the unpreconditioned iteration."""
    from nonstationary_multivariate_gaussian_process_amd import drivers
    rng = np.random.default_rng(1)
    P, r = 300, 6
    Q, _ = np.linalg.qr(rng.standard_normal((P, r)))
    ev = np.array([4e4, 900.0, 60.0, -7.0, 2.5, 0.2])
    A = Q @ np.diag(ev) @ Q.T + 1e-3 * np.diag(rng.standard_normal(P))
    U, lam, info = drivers._randomized_eigs(lambda V: V @ A, P, 24, 12, 1, 0.5, 0)
    assert U.shape == (5, P) and np.allclose(lam, [4e4, 900.0, 60.0, 7.0, 2.5], rtol=1e-3)          # 0.2 dropped, -7 kept as 7
    assert info["negative_kept"] and abs(info["most_negative"] + 7.0) < 0.05 and np.allclose(U @ U.T, np.eye(5), atol=1e-10)
    assert np.allclose(np.abs(U @ Q[:, :5]), np.eye(5), atol=1e-3)

    class Met:
        def __init__(self, U, lam):
            self.U, self.lam, self.info = U, lam, {"most_negative": 0.0}
            self.rank = 0 if U is None else U.shape[0]
    lam_pos = np.array([4e4, 900.0, 60.0, 7.0, 2.5])
    Hm = np.eye(P) + Q[:, :5] @ np.diag(lam_pos) @ Q[:, :5].T
    b = rng.standard_normal(P)

    def f(w, to_pars):
        x = to_pars(w)
        return 0.5 * x @ Hm @ x - b @ x + 0.05 * np.sum(x ** 4), Hm @ x - b + 0.2 * x ** 3
    out = {}
    for tag, met in (("preconditioned", Met(Q[:, :5].T.copy(), lam_pos)), ("plain", Met(None, None))):
        nev = [0]

        def fc(w, to_pars):
            nev[0] += 1
            return f(w, to_pars)
        q, fv, gn, n = drivers._preconditioned_lbfgs(fc, lambda q0: (lambda w: q0 + w), lambda q_at, rnd: met, np.zeros(P), 400, 1, 30, 1e-9,
                                                     None, nev)
        out[tag] = (fv, gn, n)
        assert gn < 1e-6
    assert abs(out["preconditioned"][0] - out["plain"][0]) < 1e-8
    assert out["preconditioned"][2] < out["plain"][2]          # (96 against 142 evaluations here; at N = 2048 it is 57 against > 2,000)
