"""Empirical initialiser of the nonseparable model's curves.

Host-side NumPy/SciPy, not on the GPU path: a one-off O(N W^2 M) computation that supplies the starting point of the MAP loop
(the reference calls it once per subject before ``train()``, ``Nonseparable_model_mpisim.py:315``).  Restates
``Utility/empirical_estimation.py:36-133``: for every location a window of neighbours, per output an experimental
semivariogram over all pairs of the window fitted with the Gaussian variogram ``sigma^2 (1 - exp(-s^2 / (2 l^2)))`` (SciPy
``curve_fit``, as the reference), the window's second-moment matrix and its Cholesky factor, and a moving average of the
fitted length-scales.  Pairs are generated in the reference's order (i < j, row-major) so that the least-squares problems are
the same arrays; the pair loops are vectorised.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import curve_fit

PRECISION = 1e-6      # Utility/settings.py:6


def variogram_gaussian(s, sigma, ell):
    """Gaussian variogram, empirical_estimation.py:59-60."""
    return sigma ** 2 * (1.0 - np.exp(-0.5 * s ** 2 / ell ** 2))


def semivariogram(x, y):
    """Experimental semivariogram of one output over all pairs i < j: (lags, 0.5 (y_j - y_i)^2); empirical_estimation.py:36-56."""
    i, j = np.triu_indices(x.shape[0], 1)
    d = y[j] - y[i]
    # The reference squares NumPy SCALARS (`(a - b)**2` -> libm's scalar pow, < 1 ulp but not correctly rounded); NumPy's array
    # paths (x*x, or the SIMD pow) differ from it by one ulp now and then, which is enough to move a flat least-squares fit by
    # 1e-7.  The squares therefore go through the same scalar pow; the fits, not this loop, dominate the cost.
    sq = np.fromiter((math.pow(v, 2.0) for v in d), dtype=np.float64, count=d.shape[0])
    return x[j] - x[i], 0.5 * sq


def lowtriangle2vec(L):
    r, c = np.tril_indices(L.shape[0])
    return L[r, c]


def global_estimation(x, Y):
    """Sample covariance of the outputs and its packed Cholesky factor; empirical_estimation.py:63-68."""
    S = np.cov(np.asarray(Y, dtype=np.float64).T)
    return S, lowtriangle2vec(np.linalg.cholesky(S))


def local_estimation(x, Y, window_size=30, smooth_window=10):
    """Windowed estimates at every location; empirical_estimation.py:71-133 (same return tuple):
    (est_sigmas [N], est_ls [N], smooth_ls [N], est_stds [N, M], est_R [N, M, M], est_B [N, M, M], est_L_vecs [N T],
    est_tilde_sigma2_err = -4).  The window of location n is x[max(0, n - W) : min(n + W, N - 1)] (end exclusive, as there)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    Y = np.asarray(Y, dtype=np.float64)
    N, M = Y.shape
    est_sigmas, est_ls = np.zeros(N), np.zeros(N)
    est_B, est_R = np.zeros((N, M, M)), np.zeros((N, M, M))
    est_stds = np.zeros((N, M))
    L_vecs = []
    for n in range(N):
        a, b = max(0, n - window_size), min(n + window_size, N - 1)
        xs, Ys = x[a:b], Y[a:b]
        cofs = np.zeros((M, 2))
        for m in range(M):
            lag, sv = semivariogram(xs, Ys[:, m])
            cofs[m], _ = curve_fit(variogram_gaussian, lag, sv, maxfev=2000)
        cof = cofs.mean(0)
        est_sigmas[n], est_ls[n] = abs(cof[0]), abs(cof[1])
        S = Ys.T @ Ys / (Ys.shape[0] - 1)
        try:
            Lf = np.linalg.cholesky(S)
        except np.linalg.LinAlgError:
            S = S + PRECISION * np.eye(M)
            Lf = np.linalg.cholesky(S)
        est_B[n] = S
        L_vecs.append(lowtriangle2vec(Lf))
        D = np.sqrt(np.diag(S))
        est_stds[n] = D
        est_R[n] = S / np.outer(D, D)
    smooth_ls = np.array([est_ls[max(0, n - smooth_window):min(n + smooth_window, N - 1)].mean() for n in range(N)])
    return est_sigmas, est_ls, smooth_ls, est_stds, est_R, est_B, np.concatenate(L_vecs), -4


def initial_parameters_svc(x, Y, window_size=30):
    """Flat nonseparable parameter vector [tilde_l | uL_vecs | tilde_sigma2_err] from the windowed estimates: log of the
    smoothed length-scales, log on the diagonal slots of the windowed Cholesky factors (what the mpisim script feeds its MAP
    loop, Nonseparable_model_mpisim.py:315-327)."""
    _, _, smooth_ls, _, _, _, L_vecs, tse = local_estimation(x, Y, window_size)
    N, M = np.asarray(Y).shape
    T = M * (M + 1) // 2
    uL = L_vecs.reshape(N, T).copy()
    d = np.cumsum(np.arange(1, M + 1)) - 1
    uL[:, d] = np.log(uL[:, d])
    return np.concatenate([np.log(smooth_ls), uL.reshape(-1), [float(tse)]])
