"""Deterministic (MAP) prediction behind the reference's signatures (reference: Utility/prediction.py).

In scope: ``point_predmap_inhomogeneous`` (:912-988), ``pointwise_`` /
``test_predmap_inhomogeneous`` (:990-1036), ``point_predmap`` / ``pointwise_predmap`` / ``test_predmap`` (:337-458),
``pointwise_predmap_S`` / ``test_predmap_S`` (:1566-1638).  The reference rebuilds and eigendecomposes the full
MN x MN covariance for every grid point; here all grid points share ONE Cholesky factor on the GPU and the
cross-covariances are solved as one multi-right-hand-side triangular solve (nmgp_predict_*).
The stochastic ``*_sampling`` / ``predsample*`` and the Hadamard variants are out of scope (no parity target).
"""
import numpy as np
import torch

from . import settings
from ._bridge import ctx, to_np, to_t


def _f(v):
    return float(v.detach()) if isinstance(v, torch.Tensor) else float(v)


def _percentiles(mean, var):
    sd = np.sqrt(var)
    return np.stack([mean - 1.96 * sd, mean, mean + 1.96 * sd], axis=1)          # [S, 3, M]


def _svc(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, xs, hyper):
    c = ctx()
    c.set_data(x, Y)
    pars = np.concatenate([to_np(tilde_l).reshape(-1), to_np(uL_vecs).reshape(-1), to_np(tilde_sigma2_err).reshape(-1)])
    mean, var, Ls = c.predict_svc(pars, hyper, to_np(xs).reshape(-1))
    return to_t(_percentiles(mean, var)), to_t(Ls)


def point_predmap_inhomogeneous(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, x_star, mu_tilde_l, alpha_tilde_l,
                                beta_tilde_l, mu_L, alpha_L, beta_L, *args, **kwargs):
    """Nonseparable model: [mu-1.96 s, mu, mu+1.96 s] ([3, M]) and the predicted L_vec ([T]) at x_star;
    reference prediction.py:912-988."""
    hyper = [_f(mu_tilde_l), _f(alpha_tilde_l), _f(beta_tilde_l), _f(mu_L), _f(alpha_L), _f(beta_L), 1.0, 1.0]
    pct, Ls = _svc(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, x_star, hyper)
    return pct[0], Ls[0]


def pointwise_predmap_inhomogeneous(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, grids, mu_tilde_l, alpha_tilde_l,
                                    beta_tilde_l, mu_L, alpha_L, beta_L, *args, **kwargs):
    """All grid points at once ([G, 3, M], [G, T]); reference prediction.py:990-1012."""
    hyper = [_f(mu_tilde_l), _f(alpha_tilde_l), _f(beta_tilde_l), _f(mu_L), _f(alpha_L), _f(beta_L), 1.0, 1.0]
    return _svc(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, grids, hyper)


def test_predmap_inhomogeneous(tilde_l, L_vecs, tilde_sigma2_err, Y, x, x_test, mu_tilde_l, alpha_tilde_l, beta_tilde_l,
                               mu_L, alpha_L, beta_L, *args, **kwargs):
    """reference prediction.py:1014-1036 (same computation on test inputs)."""
    return pointwise_predmap_inhomogeneous(tilde_l, L_vecs, tilde_sigma2_err, Y, x, x_test, mu_tilde_l, alpha_tilde_l,
                                           beta_tilde_l, mu_L, alpha_L, beta_L)


def _sep(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, xs, hyper):
    c = ctx()
    c.set_data(x, Y)
    pars = np.concatenate([to_np(tilde_l).reshape(-1), to_np(tilde_sigma).reshape(-1), to_np(uL_vec).reshape(-1),
                           to_np(tilde_sigma2_err).reshape(-1)])
    mean, var = c.predict_sep(pars, hyper, to_np(xs).reshape(-1))
    return to_t(_percentiles(mean, var))


def point_predmap(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_star, mu_tilde_l, alpha_tilde_l, beta_tilde_l,
                  mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma, *args, **kwargs):
    """Separable model, one new input ([3, M]); reference prediction.py:337-408."""
    hyper = [_f(mu_tilde_l), _f(alpha_tilde_l), _f(beta_tilde_l), _f(mu_tilde_sigma), _f(alpha_tilde_sigma),
             _f(beta_tilde_sigma), 1.0, 1.0, 10.0]
    return _sep(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_star, hyper)[0]


def pointwise_predmap(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, grids, mu_tilde_l, alpha_tilde_l,
                      beta_tilde_l, mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma, *args, **kwargs):
    """[G, 3, M]; reference prediction.py:410-430."""
    hyper = [_f(mu_tilde_l), _f(alpha_tilde_l), _f(beta_tilde_l), _f(mu_tilde_sigma), _f(alpha_tilde_sigma),
             _f(beta_tilde_sigma), 1.0, 1.0, 10.0]
    return _sep(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, grids, hyper)


def test_predmap(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_test, mu_tilde_l, alpha_tilde_l, beta_tilde_l,
                 mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma, *args, **kwargs):
    """reference prediction.py:432-458."""
    return pointwise_predmap(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_test, mu_tilde_l, alpha_tilde_l,
                             beta_tilde_l, mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma)


def _sta(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, xs):
    c = ctx()
    c.set_data(x, Y)
    pars = np.concatenate([to_np(tilde_l).reshape(-1), to_np(tilde_sigma).reshape(-1), to_np(uL_vec).reshape(-1),
                           to_np(tilde_sigma2_err).reshape(-1)])
    return c.predict_sta(pars, to_np(xs).reshape(-1))


def pointwise_predmap_S(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, grids, *args, **kwargs):
    """Stationary model on a grid ([G, 3, M]); reference prediction.py:1566-1599."""
    mean, var = _sta(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, grids)
    return to_t(_percentiles(mean, var))


def test_predmap_S(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, test_x, *args, **kwargs):
    """Stationary model: (mean [S, M], std [S, M]); reference prediction.py:1601-1638."""
    mean, var = _sta(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, test_x)
    return to_t(mean), to_t(np.sqrt(var))


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
