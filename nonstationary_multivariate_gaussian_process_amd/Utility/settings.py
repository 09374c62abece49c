"""Constants read by the path (reference: Utility/settings.py:3-6)."""
import torch

jitter = 1e-6
torchType = torch.DoubleTensor
precision = 1e-6


def __getattr__(name):
    """Names outside the mirrored path come from the user's reference checkout (Utility/_overlay.py)."""
    from . import _overlay
    return _overlay.module_getattr(__name__, name)
