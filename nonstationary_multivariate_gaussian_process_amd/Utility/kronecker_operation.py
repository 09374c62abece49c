"""Kronecker-structured operations on the GPU (reference: Utility/kronecker_operation.py).

Forward values come from the MI355X kernels; when an input requires grad the backward pass differentiates the host
restatements below (`_host_*`, plain torch), as the reference's own torch ops would."""
import torch

from ._bridge import ctx, to_np, to_t, scalar, with_host_backward


def _host_kron_spectrum(sigma2, B, K):
    wB, vB = torch.linalg.eigh(B)
    wK, vK = torch.linalg.eigh(K)
    return torch.kron(wB, wK) + sigma2, torch.kron(vB, vK)


def _host_kron_inv(sigma2, B, K):
    w, U = _host_kron_spectrum(sigma2, B, K)
    return (U / w) @ U.T


def _host_kron_logdet(sigma2, B, K):
    return torch.log(_host_kron_spectrum(sigma2, B, K)[0]).sum()


def _host_kron_mv(B, K, y):
    Y = y.reshape(B.shape[1], K.shape[1]).T              # y = vec(Y), column m of Y = block m
    return (K @ Y @ B.T).T.reshape(-1)


def kronecker_product(t1, t2):
    """t1 kron t2; reference kronecker_operation.py:5-22."""
    return with_host_backward(to_t(ctx().kron_product(to_np(t1), to_np(t2))), torch.kron, t1, t2)


def kronecker_product_diag(d1, d2):
    """diag(D1) kron diag(D2) as a vector; reference kronecker_operation.py:25-33."""
    val = to_t(ctx().kron_product(to_np(d1).reshape(-1, 1), to_np(d2).reshape(-1, 1)).reshape(-1))
    return with_host_backward(val, lambda a, b: torch.kron(a.reshape(-1), b.reshape(-1)), d1, d2)


def kron_inv(sigma2, B, K):
    """(sigma2 I + B kron K)^-1 through the two small eigendecompositions; reference kronecker_operation.py:36-54."""
    inv, _ = ctx().kron_inv_logdet(scalar(sigma2), to_np(B), to_np(K), want_inv=True)
    return with_host_backward(to_t(inv), _host_kron_inv, sigma2, B, K)


def kron_logdet(sigma2, B, K):
    """log det(sigma2 I + B kron K); reference kronecker_operation.py:57-69."""
    _, ld = ctx().kron_inv_logdet(scalar(sigma2), to_np(B), to_np(K), want_inv=False)
    return with_host_backward(torch.tensor(float(ld), dtype=torch.float64), _host_kron_logdet, sigma2, B, K)


def kron_mv(B, K, y):
    """(B kron K) y = vec(K Y B^T) without forming the product; reference kronecker_operation.py:72-85."""
    return with_host_backward(to_t(ctx().kron_mv(to_np(B), to_np(K), to_np(y))), _host_kron_mv, B, K, y)


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
