"""Tensor <-> C-ABI plumbing shared by the Utility mirror modules."""
import numpy as np
import torch

from .. import _lib


def ctx():
    return _lib.default_context()


def no_grad_inputs(fn_name, *tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            raise NotImplementedError(
                "%s: gradients do not flow through this primitive on the MI355X path; differentiate through "
                "logpos.nlogpos_obj / nlogpos_obj_SVC / nlogpos_obj_S (fused value+gradient) instead, or call it "
                "with detached tensors." % fn_name)


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
