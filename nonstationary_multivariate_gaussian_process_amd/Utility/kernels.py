"""Covariance kernels on the GPU behind the reference signatures (reference: Utility/kernels.py)."""
import torch

from . import settings  # noqa: F401
from ._bridge import ctx, no_grad_inputs, to_np, to_t, scalar


def pairwise_distances(x, y=None):
    """dist[i,j] = ||x_i||^2 + ||y_j||^2 - 2 x_i.y_j (expanded form); reference kernels.py:5-21."""
    no_grad_inputs("pairwise_distances", x, y)
    return to_t(ctx().pairwise_distances(to_np(x), to_np(y)))


def RBF_cov(X1, X2=None, alpha=1., beta=1.):
    """alpha^2 exp(-0.5 ||x/beta - x'/beta||^2) (+ jitter I when X2 is None); reference kernels.py:24-43."""
    no_grad_inputs("RBF_cov", X1, X2, alpha, beta)
    return to_t(ctx().rbf_cov(to_np(X1), to_np(X2), scalar(alpha), scalar(beta)))


def Nonstationary_RBF_cov(X1, sigma1=None, ell1=None, X2=None, sigma2=None, ell2=None):
    """Gibbs kernel with per-input scale sigma and length-scale ell; reference kernels.py:46-73."""
    no_grad_inputs("Nonstationary_RBF_cov", X1, sigma1, ell1, X2, sigma2, ell2)
    if X2 is None:
        sigma2 = ell2 = None           # the reference overwrites them with sigma1 / ell1 (kernels.py:61-62)
    elif sigma2 is None or ell2 is None:
        # the reference would fail on None.view(); keep the failure loud
        raise TypeError("Nonstationary_RBF_cov: sigma2 and ell2 are required when X2 is given")
    return to_t(ctx().nonstat_rbf_cov(to_np(X1), to_np(sigma1), to_np(ell1), to_np(X2), to_np(sigma2), to_np(ell2)))


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
