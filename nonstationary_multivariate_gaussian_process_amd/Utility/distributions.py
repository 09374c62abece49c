"""Log densities used by the objectives (reference: Utility/distributions.py)."""
import numpy as np
import torch
from scipy.special import gammaln as lgamma

from . import settings
from ._bridge import ctx, no_grad_inputs, to_np, scalar


def _dt(v):
    return torch.tensor(float(v), dtype=torch.float64)


def multivariate_normal_logpdf(y, mu, logdetSigma, invSigma):
    """-0.5 logdet - 0.5 (y-mu)' invSigma (y-mu): the 2 pi term is dropped (reference distributions.py:21-22)."""
    no_grad_inputs("multivariate_normal_logpdf", y, mu, logdetSigma, invSigma)
    return _dt(ctx().mvn_logpdf(to_np(y), to_np(mu), scalar(logdetSigma), to_np(invSigma)))


def multivariate_normal_logpdf0(y, mu, B, K, sigma2):
    """Density for covariance B kron K + sigma2 I in the joint eigenbasis; reference distributions.py:26-52."""
    no_grad_inputs("multivariate_normal_logpdf0", y, mu, B, K, sigma2)
    return _dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), to_np(B), to_np(K), scalar(sigma2)))


def multivariate_normal_logpdf1(y, mu, B, K, sigma2):
    """Robust variant (reference distributions.py:55-96: random diagonal jitter of size `precision` on B and K
    before the eigendecompositions).  The jitter draw uses torch's RNG exactly as the reference does."""
    no_grad_inputs("multivariate_normal_logpdf1", y, mu, B, K, sigma2)
    Bj = to_np(B) + np.diag(torch.rand(B.size(0)).type(settings.torchType).numpy() * settings.precision)
    Kj = to_np(K) + np.diag(torch.rand(K.size(0)).type(settings.torchType).numpy() * settings.precision)
    return _dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), Bj, Kj, scalar(sigma2)))


def multivariate_normal_logpdf2(y, mu, B, K, sigma2):
    """Dense evaluation of the same density; reference distributions.py:99-113."""
    no_grad_inputs("multivariate_normal_logpdf2", y, mu, B, K, sigma2)
    return _dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), to_np(B), to_np(K), scalar(sigma2), dense=True))


def inverse_gamma_logpdf_u(x, alpha=1., beta=1.):
    """Un-normalised inverse-gamma log density (scalar, host); reference distributions.py:116-124."""
    return (-alpha - 1) * torch.log(x) - beta / x


def inverse_gamma_logpdf(x, alpha=1., beta=1.):
    """Normalised inverse-gamma log density (scalar, host); reference distributions.py:126-134."""
    return (-alpha - 1) * torch.log(x) - beta / x + alpha * np.log(beta) - lgamma(alpha)


def gamma_logpdf(x, alpha=1., beta=1.):
    """reference distributions.py:136-137."""
    return (alpha - 1) * torch.log(x) - beta * x + alpha * np.log(beta) - lgamma(alpha)


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
