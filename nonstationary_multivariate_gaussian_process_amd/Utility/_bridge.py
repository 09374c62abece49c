"""Tensor <-> C-ABI plumbing shared by the Utility mirror modules."""
import numpy as np
import torch

from .. import _lib


def ctx():
    return _lib.default_context()


def _wants_grad(tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class _GpuForwardHostBackward(torch.autograd.Function):
    """Forward: the MI355X primitive (already evaluated: `value`).  Backward: the same expression restated with torch ops on
    the host (`host_expr`), differentiated by autograd -- the reference's primitives are plain torch ops that gradients flow
    through (kernels.py:46-73, kronecker_operation.py:5-85); here the fused objectives carry the production gradient, and
    this keeps a user's own composition of the primitives differentiable at the host's speed."""

    @staticmethod
    def forward(actx, value, host_expr, *inputs):
        actx.host_expr = host_expr
        actx.inputs = inputs
        return value

    @staticmethod
    def backward(actx, gout):
        ins = [t.detach().clone().requires_grad_(True) if isinstance(t, torch.Tensor) and t.requires_grad else
               (t.detach() if isinstance(t, torch.Tensor) else t) for t in actx.inputs]
        with torch.enable_grad():
            out = actx.host_expr(*ins)
            wrt = [t for t in ins if isinstance(t, torch.Tensor) and t.requires_grad]
            gs = list(torch.autograd.grad(out, wrt, gout, allow_unused=True))
        res = []
        for t in ins:
            res.append(gs.pop(0) if isinstance(t, torch.Tensor) and t.requires_grad else None)
        return (None, None, *res)


def with_host_backward(value, host_expr, *inputs):
    """`value` (tensor computed on the GPU from `inputs`) as a differentiable function of the inputs that require grad."""
    if not _wants_grad(inputs):
        return value
    return _GpuForwardHostBackward.apply(value, host_expr, *inputs)


def to_np(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float64, copy=False))
    return np.ascontiguousarray(np.asarray(t, dtype=np.float64))


def to_t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).type(torch.DoubleTensor)


def scalar(v):
    if isinstance(v, torch.Tensor):
        return float(v.detach())
    return float(v)
