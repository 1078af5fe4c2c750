"""Small helpers shared by the device-backed pylib modules."""
import numpy as np
import torch

from .. import _lib
from .._lib import lib, check, ptr, stream, require_gpu


def dev():
    return _lib.device()


def to_dev(x, dtype):
    """numpy / list / CPU or GPU tensor -> contiguous GPU tensor of `dtype`."""
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=dev(), dtype=dtype).contiguous()
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(dev()).contiguous()


__all__ = ['lib', 'check', 'ptr', 'stream', 'dev', 'to_dev', 'np', 'torch']
