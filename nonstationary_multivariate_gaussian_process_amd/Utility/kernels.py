"""Covariance kernels on the GPU behind the reference signatures (reference: Utility/kernels.py).

Forward values come from the MI355X kernels (k_rect); when an input requires grad the backward pass differentiates the host
restatements below (`_host_*`, plain torch), as the reference's own torch ops would."""
import torch

from . import settings
from ._bridge import ctx, to_np, to_t, scalar, with_host_backward


def _host_sqdist(x, y=None):
    y = x if y is None else y
    return (x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2.0 * (x @ y.T)


def _host_rbf(X1, X2, alpha, beta):
    K = torch.exp(-0.5 * _host_sqdist(X1 / beta, None if X2 is None else X2 / beta)) * alpha ** 2
    return K + settings.jitter * torch.eye(X1.shape[0], dtype=K.dtype) if X2 is None else K


def _host_gibbs(X1, sigma1, ell1, X2, sigma2, ell2):
    one = lambda n: torch.ones(n, dtype=torch.float64)      # noqa: E731
    sigma1 = one(X1.shape[0]) if sigma1 is None else sigma1
    ell1 = one(X1.shape[0]) if ell1 is None else ell1
    same = X2 is None
    if same:
        X2, sigma2, ell2 = X1, sigma1, ell1
    A = (ell1 ** 2)[:, None] + (ell2 ** 2)[None, :]
    K = (sigma1[:, None] * sigma2[None, :]) * torch.sqrt(2.0 * (ell1[:, None] * ell2[None, :]) / A) * torch.exp(-_host_sqdist(X1, X2) / A)
    return K + settings.jitter * torch.eye(X1.shape[0], dtype=K.dtype) if same else K


def pairwise_distances(x, y=None):
    """dist[i,j] = ||x_i||^2 + ||y_j||^2 - 2 x_i.y_j (expanded form); reference kernels.py:5-21."""
    return with_host_backward(to_t(ctx().pairwise_distances(to_np(x), to_np(y))), _host_sqdist, x, y)


def RBF_cov(X1, X2=None, alpha=1., beta=1.):
    """alpha^2 exp(-0.5 ||x/beta - x'/beta||^2) (+ jitter I when X2 is None); reference kernels.py:24-43."""
    return with_host_backward(to_t(ctx().rbf_cov(to_np(X1), to_np(X2), scalar(alpha), scalar(beta))), _host_rbf, X1, X2, alpha, beta)


def Nonstationary_RBF_cov(X1, sigma1=None, ell1=None, X2=None, sigma2=None, ell2=None):
    """Gibbs kernel with per-input scale sigma and length-scale ell; reference kernels.py:46-73."""
    if X2 is None:
        sigma2 = ell2 = None           # the reference overwrites them with sigma1 / ell1 (kernels.py:61-62)
    elif sigma2 is None or ell2 is None:
        # the reference would fail on None.view(); keep the failure loud
        raise TypeError("Nonstationary_RBF_cov: sigma2 and ell2 are required when X2 is given")
    val = to_t(ctx().nonstat_rbf_cov(to_np(X1), to_np(sigma1), to_np(ell1), to_np(X2), to_np(sigma2), to_np(ell2)))
    return with_host_backward(val, _host_gibbs, X1, sigma1, ell1, X2, sigma2, ell2)


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
