"""Synthetic inputs for the log-posterior path (host side, NumPy only).

Follows the data recipe of the reference simulator (``SIM_code/sim.py:173-275`` ``SIM_MNTS``) generalised
from its hard-wired M=2 to M outputs as fixed in SURVEY.md section 8(d):

    x          = sort(U(0,1)^N)                                      sim.py:177
    tilde_l    = 3 (x-1)^3 - 3                                       sim.py:180
    s_m(x)     = 1 + x^2 (m even),  2 - x^2 (m odd)                  sim.py:220
    R_mm'(x)   = clip(cos(pi x), +-0.95)^{|m-m'|}                    sim.py:233,243-245 (M=2 case: cos(pi x))
    B(x)       = D R D,  L(x) = chol(B(x))                           sim.py:241-248
    sigma2_err = 1e-2                                                sim.py:254
    y ~ N(0, K + sigma2 I),  Y = y.view(M, N)^T                      sim.py:256-263

plus the deterministic "RNG-free" curves used for the committed known-answer fixtures.
Nothing here touches the GPU; the arrays it returns are what ``nmgp_set_data`` / ``nlogpos_obj_*`` consume.
"""
from __future__ import annotations

import numpy as np

JITTER = 1e-6


def tril_diag_slots(M):
    return np.cumsum(np.arange(1, M + 1)) - 1


def _gibbs(x, ell):
    d = x[:, None] ** 2 + x[None, :] ** 2 - 2.0 * np.outer(x, x)
    A = ell[:, None] ** 2 + ell[None, :] ** 2
    return np.sqrt(2.0 * np.outer(ell, ell) / A) * np.exp(-d / A) + JITTER * np.eye(x.shape[0])


def svc_truth(x, M):
    """True (tilde_l [N], uL [N,T]) curves of the nonseparable generator at inputs x."""
    N = x.shape[0]
    T = M * (M + 1) // 2
    tilde_l = 3.0 * (x - 1.0) ** 3 - 3.0
    s = np.stack([(1.0 + x ** 2) if m % 2 == 0 else (2.0 - x ** 2) for m in range(M)], 1)      # [N, M]
    rho = np.clip(np.cos(np.pi * x), -0.95, 0.95)
    expo = np.abs(np.arange(M)[:, None] - np.arange(M)[None, :])
    R = rho[:, None, None] ** expo[None, :, :]
    B = s[:, :, None] * R * s[:, None, :]
    L = np.linalg.cholesky(B)                                                                    # [N, M, M]
    r, c = np.tril_indices(M)
    Lv = L[:, r, c]
    uL = Lv.copy()
    d = tril_diag_slots(M)
    uL[:, d] = np.log(uL[:, d])
    assert uL.shape == (N, T)
    return tilde_l, uL, L


def simulate_nonseparable(N, M, seed, sigma2_err=1e-2):
    """One synthetic subject of the nonseparable model.  Returns dict(x, Y, pars_true, tilde_l, uL, sigma2_err)."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.random(N))
    tilde_l, uL, L = svc_truth(x, M)
    Kx = _gibbs(x, np.exp(tilde_l))
    # K[(m,i),(m',j)] = Kx[i,j] (L_i L_j^T)[m,m']
    Lcat = L.transpose(1, 0, 2).reshape(M * N, M)            # output-major rows (m, i)
    K = (Lcat @ Lcat.T) * np.tile(Kx, (M, M))
    K[np.diag_indices(M * N)] += sigma2_err
    y = np.linalg.cholesky(K) @ rng.standard_normal(M * N)
    Y = np.ascontiguousarray(y.reshape(M, N).T)
    pars = np.concatenate([tilde_l, uL.reshape(-1), [np.log(sigma2_err)]])
    return dict(x=x, Y=Y, pars_true=pars, tilde_l=tilde_l, uL=uL, sigma2_err=sigma2_err)


def simulate_separable(N, M, seed, sigma2_err=1e-2):
    """Separable generator: constant cross-output factor, nonstationary length-scale and scale curves."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.random(N))
    tilde_l = 3.0 * (x - 1.0) ** 3 - 3.0
    tilde_sigma = 0.3 * np.sin(3.0 * x)
    T = M * (M + 1) // 2
    expo = np.abs(np.arange(M)[:, None] - np.arange(M)[None, :])
    B = (0.5 ** expo) * np.outer(1.0 + 0.25 * np.arange(M), 1.0 + 0.25 * np.arange(M))
    L = np.linalg.cholesky(B)
    r, c = np.tril_indices(M)
    uL = L[r, c].copy()
    d = tril_diag_slots(M)
    uL[d] = np.log(uL[d])
    sig = np.exp(tilde_sigma)
    Kx = np.outer(sig, sig) * (_gibbs(x, np.exp(tilde_l)) - JITTER * np.eye(N)) + JITTER * np.eye(N)
    Lk = np.linalg.cholesky(Kx + 1e-10 * np.eye(N))
    Z = rng.standard_normal((N, M))
    Yf = Lk @ Z @ L.T                                          # vec(Yf) ~ N(0, B kron Kx)
    Y = Yf + np.sqrt(sigma2_err) * rng.standard_normal((N, M))
    pars = np.concatenate([tilde_l, tilde_sigma, uL, [np.log(sigma2_err)]])
    assert pars.shape[0] == 2 * N + T + 1
    return dict(x=x, Y=np.ascontiguousarray(Y), pars_true=pars, sigma2_err=sigma2_err)


def simulate_stationary(N, M, seed, tilde_l=-1.0, tilde_sigma=1.0, sigma2_err=1e-2):
    """Stationary LMC generator (recipe of sim.py:81-100: RBF with log-lengthscale -1, log-scale 1)."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.random(N))
    Lm = rng.standard_normal((M, M))
    Bm = Lm @ Lm.T + 0.1 * np.eye(M)
    L = np.linalg.cholesky(Bm)
    r, c = np.tril_indices(M)
    uL = L[r, c].copy()
    d = tril_diag_slots(M)
    uL[d] = np.log(uL[d])
    ell = np.exp(tilde_l) * np.ones(N)
    Kx = np.exp(2.0 * tilde_sigma) * (_gibbs(x, ell) - JITTER * np.eye(N)) + JITTER * np.eye(N)
    Lk = np.linalg.cholesky(Kx + 1e-10 * np.eye(N))
    Y = Lk @ rng.standard_normal((N, M)) @ L.T + np.sqrt(sigma2_err) * rng.standard_normal((N, M))
    pars = np.concatenate([[tilde_l, tilde_sigma], uL, [np.log(sigma2_err)]])
    return dict(x=x, Y=np.ascontiguousarray(Y), pars_true=pars, sigma2_err=sigma2_err)


def perturb(pars, scale=0.05, phase=0.0):
    """Smooth deterministic perturbation of a parameter vector (evaluation away from the truth)."""
    k = np.arange(pars.shape[0], dtype=np.float64)
    return pars + scale * np.sin(0.01 * k + phase)


# ---- RNG-free known-answer inputs ---------------------------------------
def rngfree_inputs(N, M):
    x = np.linspace(0.05, 0.95, N)
    Y = np.stack([np.sin(2.0 * np.pi * x * (m + 1)) + 0.1 * m for m in range(M)], 1)
    return x, np.ascontiguousarray(Y)


def rngfree_pars_svc(N, M):
    x, _ = rngfree_inputs(N, M)
    T = M * (M + 1) // 2
    tl = 3.0 * (x - 1.0) ** 3 - 3.0
    uL = np.stack([0.1 * (t + 1) * np.cos(np.pi * x) - 0.2 for t in range(T)], 1)
    return np.concatenate([tl, uL.reshape(-1), [np.log(1e-2)]])


def rngfree_pars_sep(N, M):
    x, _ = rngfree_inputs(N, M)
    T = M * (M + 1) // 2
    tl = 3.0 * (x - 1.0) ** 3 - 3.0
    ts = 0.3 * np.sin(3.0 * x)
    uL = 0.1 * (np.arange(T) + 1.0) - 0.2
    return np.concatenate([tl, ts, uL, [np.log(1e-2)]])


def rngfree_pars_sta(M):
    T = M * (M + 1) // 2
    uL = 0.1 * (np.arange(T) + 1.0) - 0.2
    return np.concatenate([[-2.0, 0.0], uL, [np.log(1e-2)]])


# Hyper-parameter sets seen in the reference's callers.
HYPER_SVC = dict(mu_tilde_l=0.0, alpha_tilde_l=10.0, beta_tilde_l=1.0, mu_L=0.0, alpha_L=10.0, beta_L=1.0, a=1.0, b=1.0)
HYPER_SVC_MPISIM = dict(mu_tilde_l=0.0, alpha_tilde_l=10.0, beta_tilde_l=1.0, mu_L=0.0, alpha_L=1.0, beta_L=1.0,
                        a=1e-2, b=1e-2)
HYPER_SVC_DIST = dict(mu_tilde_l=0.0, alpha_tilde_l=5.0, beta_tilde_l=0.1, mu_L=0.0, alpha_L=5.0, beta_L=0.2,
                      a=1.0, b=1.0)
HYPER_SEP = dict(mu_tilde_l=0.0, alpha_tilde_l=10.0, beta_tilde_l=1.0, mu_tilde_sigma=0.0, alpha_tilde_sigma=10.0,
                 beta_tilde_sigma=1.0, a=1.0, b=1.0, c=10.0)
HYPER_STA = dict(mu_tilde_l=0.0, sigma_tilde_l=1.0, a=1.0, b=1.0, c=10.0)
