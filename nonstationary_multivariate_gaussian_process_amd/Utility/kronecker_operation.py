"""Kronecker-structured operations on the GPU (reference: Utility/kronecker_operation.py)."""
import torch

from ._bridge import ctx, no_grad_inputs, to_np, to_t, scalar


def kronecker_product(t1, t2):
    """t1 kron t2; reference kronecker_operation.py:5-22."""
    no_grad_inputs("kronecker_product", t1, t2)
    return to_t(ctx().kron_product(to_np(t1), to_np(t2)))


def kronecker_product_diag(d1, d2):
    """diag(D1) kron diag(D2) as a vector; reference kronecker_operation.py:25-33."""
    no_grad_inputs("kronecker_product_diag", d1, d2)
    return to_t(ctx().kron_product(to_np(d1).reshape(-1, 1), to_np(d2).reshape(-1, 1)).reshape(-1))


def kron_inv(sigma2, B, K):
    """(sigma2 I + B kron K)^-1 through the two small eigendecompositions; reference kronecker_operation.py:36-54."""
    no_grad_inputs("kron_inv", sigma2, B, K)
    inv, _ = ctx().kron_inv_logdet(scalar(sigma2), to_np(B), to_np(K), want_inv=True)
    return to_t(inv)


def kron_logdet(sigma2, B, K):
    """log det(sigma2 I + B kron K); reference kronecker_operation.py:57-69."""
    no_grad_inputs("kron_logdet", sigma2, B, K)
    _, ld = ctx().kron_inv_logdet(scalar(sigma2), to_np(B), to_np(K), want_inv=False)
    return torch.tensor(float(ld), dtype=torch.float64)


def kron_mv(B, K, y):
    """(B kron K) y = vec(K Y B^T) without forming the product; reference kronecker_operation.py:72-85."""
    no_grad_inputs("kron_mv", B, K, y)
    return to_t(ctx().kron_mv(to_np(B), to_np(K), to_np(y)))


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
