"""The helpers of utils/miscellaneous.py that sit on the path: :37-58 (flips) and :233-247 (ramps)."""
import numpy as np
import torch


def _hflip(tensor):
    return torch.flip(tensor, (3,))


def batch_hfilp(tensor):
    if isinstance(tensor, (list, tuple)):
        return [_hflip(t) for t in tensor]
    return _hflip(tensor)


def batch_boxlist_hflip(boxlists):
    return [b.transpose(0) for b in boxlists]


def sigmoid_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    current = np.clip(current, 0.0, rampup_length)
    phase = 1.0 - current / rampup_length
    return float(np.exp(-5.0 * phase * phase))


def sigmoid_rampdown(gap_time, rampdown_length):
    if rampdown_length == 0:
        return 1.0
    phase = 1.0 - gap_time / rampdown_length
    return float(np.exp(-12 * phase * phase))
