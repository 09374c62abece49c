"""Log densities used by the objectives (reference: Utility/distributions.py).

Forward values come from the MI355X entry points; when an input requires grad the backward pass differentiates the host
restatements below (`_host_*`, plain torch), as the reference's own torch ops would."""
import numpy as np
import torch
from scipy.special import gammaln as lgamma

from . import settings
from ._bridge import ctx, to_np, scalar, with_host_backward


def _dt(v):
    return torch.tensor(float(v), dtype=torch.float64)


def _host_mvn(y, mu, logdetSigma, invSigma):
    r = y - mu
    return -0.5 * logdetSigma - 0.5 * torch.dot(r, invSigma @ r)


def _host_mvn_kron(y, mu, B, K, sigma2):
    wB, vB = torch.linalg.eigh(B)
    wK, vK = torch.linalg.eigh(K)
    R = (y - mu).reshape(B.shape[0], K.shape[0]).T          # [N, M]: column m = block m of the output-major vector
    a = (vK.T @ R @ vB).T.reshape(-1)                        # (V_B kron V_K)^T (y - mu)
    w = torch.kron(wB, wK) + sigma2
    return -0.5 * torch.log(w).sum() - 0.5 * torch.dot(a / w, a)


def multivariate_normal_logpdf(y, mu, logdetSigma, invSigma):
    """-0.5 logdet - 0.5 (y-mu)' invSigma (y-mu): the 2 pi term is dropped (reference distributions.py:21-22)."""
    return with_host_backward(_dt(ctx().mvn_logpdf(to_np(y), to_np(mu), scalar(logdetSigma), to_np(invSigma))), _host_mvn,
                              y, mu, logdetSigma, invSigma)


def multivariate_normal_logpdf0(y, mu, B, K, sigma2):
    """Density for covariance B kron K + sigma2 I in the joint eigenbasis; reference distributions.py:26-52."""
    return with_host_backward(_dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), to_np(B), to_np(K), scalar(sigma2))), _host_mvn_kron,
                              y, mu, B, K, sigma2)


def multivariate_normal_logpdf1(y, mu, B, K, sigma2):
    """Robust variant (reference distributions.py:55-96: random diagonal jitter of size `precision` on B and K
    before the eigendecompositions).  The jitter draw uses torch's RNG exactly as the reference does."""
    jB = torch.rand(B.size(0)).type(settings.torchType) * settings.precision
    jK = torch.rand(K.size(0)).type(settings.torchType) * settings.precision
    val = _dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), to_np(B) + np.diag(jB.numpy()), to_np(K) + np.diag(jK.numpy()), scalar(sigma2)))
    return with_host_backward(val, lambda y_, mu_, B_, K_, s_: _host_mvn_kron(y_, mu_, B_ + torch.diag(jB), K_ + torch.diag(jK), s_),
                              y, mu, B, K, sigma2)


def multivariate_normal_logpdf2(y, mu, B, K, sigma2):
    """Dense evaluation of the same density; reference distributions.py:99-113."""
    return with_host_backward(_dt(ctx().mvn_logpdf_kron(to_np(y), to_np(mu), to_np(B), to_np(K), scalar(sigma2), dense=True)),
                              _host_mvn_kron, y, mu, B, K, sigma2)


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
