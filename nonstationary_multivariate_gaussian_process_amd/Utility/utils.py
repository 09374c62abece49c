"""Parameter packing helpers (reference: Utility/utils.py:10-88).  Pure index arithmetic on T-length vectors:
these stay on the host (torch / NumPy, differentiable like the reference's), the GPU path unpacks parameters
itself inside the fused kernels."""
import numpy as np
import torch

from . import settings


def _slots(M):
    T = int(M * (M + 1) / 2)
    on = list(np.cumsum(np.arange(1, M + 1)) - 1)
    off = [k for k in range(T) if k not in on]
    return T, on, off


def uLvec2Lvec(uL_vec, M):
    """exp on the diagonal slots of a packed lower triangle; reference utils.py:10-22."""
    T, on, off = _slots(M)
    if isinstance(uL_vec, torch.Tensor):
        L_vec = torch.zeros_like(uL_vec)
        L_vec[on] = torch.exp(uL_vec[on])
        L_vec[off] = uL_vec[off]
    else:
        L_vec = np.zeros_like(uL_vec)
        L_vec[on] = np.exp(uL_vec[on])
        L_vec[off] = uL_vec[off]
    return L_vec


def Lvec2uLvec(L_vec, M):
    """log on the diagonal slots; reference utils.py:24-36."""
    T, on, off = _slots(M)
    if isinstance(L_vec, torch.Tensor):
        uL_vec = torch.zeros_like(L_vec)
        uL_vec[on] = torch.log(L_vec[on])
        uL_vec[off] = L_vec[off]
    else:
        uL_vec = np.zeros_like(L_vec)
        uL_vec[on] = np.log(L_vec[on])
        uL_vec[off] = L_vec[off]
    return uL_vec


def _batched(vecs, N, M, fn_t, fn_n):
    T, on, off = _slots(M)
    if isinstance(vecs, torch.Tensor):
        A = vecs.reshape(N, T)
        out = torch.zeros_like(A)
        out[:, on] = fn_t(A[:, on])
        out[:, off] = A[:, off]
        return out.reshape(-1)
    A = np.asarray(vecs).reshape(N, T)
    out = np.zeros_like(A)
    out[:, on] = fn_n(A[:, on])
    out[:, off] = A[:, off]
    return out.reshape(-1)


def uLvecs2Lvecs(uL_vecs, N, M):
    """Location-major batch of uLvec2Lvec (vectorised; the reference loops over N); reference utils.py:38-46."""
    return _batched(uL_vecs, N, M, torch.exp, np.exp)


def Lvecs2uLvecs(L_vecs, N, M):
    """reference utils.py:48-54."""
    return _batched(L_vecs, N, M, torch.log, np.log)


def vec2lowtriangle(x, N=None):
    """Packed row-major lower triangle -> dense [N, N]; reference utils.py:56-74."""
    if isinstance(x, torch.Tensor):
        if N * (N + 1) / 2 != x.size(0):
            raise ValueError("check the dimension size!")
        mat = torch.zeros([N, N]).type(settings.torchType)
        idx = torch.tril_indices(N, N)
        mat[idx[0], idx[1]] = x
        return mat
    if N * (N + 1) / 2 != x.shape[0]:
        raise ValueError("check the dimension size!")
    mat = np.zeros([N, N])
    idx = np.tril_indices(N)
    mat[idx[0], idx[1]] = x
    return mat


def lowtriangle2vec(L, N=None):
    """reference utils.py:77-88."""
    if isinstance(L, torch.Tensor):
        idx = torch.tril_indices(N, N)
        return L[idx[0], idx[1]]
    idx = np.tril_indices(N)
    return L[idx[0], idx[1]]


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
