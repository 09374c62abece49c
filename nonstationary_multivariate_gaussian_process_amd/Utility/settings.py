"""Constants read by the path (reference: Utility/settings.py:3-6)."""
import torch

jitter = 1e-6
torchType = torch.DoubleTensor
precision = 1e-6
