"""The helpers of utils/miscellaneous.py that sit on the path: :37-58 (flips) and :233-247 (ramps)."""
import threading

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


_DEV_CONST = {}
_DEV_CONST_OLD = {}   # the previous generation: kept referenced, kernels already queued on EITHER stream may still read them


def dev_const(values, dtype, device):
    """small constant tensor on the device, built once per distinct content.  `torch.tensor(list, device=cuda)` is a
    pageable host->device copy: it blocks the host until the stream has drained, which on this path (3000 launches per
    step) throws away the whole launch-ahead of the host each time (measured: 10 such calls = 22 of 56 ms host time)."""
    def freeze(v):
        return tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else v
    key = (freeze(values), dtype, str(device))
    t = _DEV_CONST.get(key)
    if t is None:
        if len(_DEV_CONST) > 4096:
            # never free on overflow: a constant may be in use by launches queued on the other stream (the teacher thread
            # shares this table), and the allocator would hand its memory to the next allocation of the creating stream
            _DEV_CONST_OLD.clear()
            _DEV_CONST_OLD.update(_DEV_CONST)
            _DEV_CONST.clear()
        t = torch.tensor(values, dtype=dtype, device=device)
        _DEV_CONST[key] = t
    return t


_INT_RING = {}


_FLOAT_RING = {}


def dev_floats(values, device):
    """dev_ints for a short fp32 vector (image bounds of `BoxList.clip_to_image`): pinned ring slot + asynchronous copy"""
    n = len(values)
    if n > 16:
        return torch.tensor(values, dtype=torch.float32).to(device, non_blocking=True)
    ent = _FLOAT_RING.get(device)
    if ent is None:
        ent = _FLOAT_RING[device] = [torch.zeros((1024, 16), dtype=torch.float32).pin_memory(), 0, threading.Lock()]
    with ent[2]:
        i = ent[1]
        ent[1] = (i + 1) & 1023
    slot = ent[0][i, :n]
    slot.copy_(torch.tensor(values, dtype=torch.float32))
    return slot.to(device, non_blocking=True)


def dev_ints(values, device):
    """a short int32 vector whose content changes from step to step (per-image proposal counts and their prefix sums) on the
    device WITHOUT draining the stream: the values are written into a slot of a pinned ring and copied asynchronously on the
    current stream (a `torch.tensor(list, device=cuda)` is a pageable copy that blocks the host, see `dev_const`, and caching by
    content does not help when the content is new every step).  A slot is re-used after 256 calls, i.e. tens of steps later."""
    n = len(values)
    if n > 64:
        return torch.tensor(values, dtype=torch.int32).to(device, non_blocking=True)
    ent = _INT_RING.get(device)
    if ent is None:
        ent = _INT_RING[device] = [torch.zeros((256, 64), dtype=torch.int32).pin_memory(), 0, threading.Lock()]
    with ent[2]:   # (the teacher thread shares the ring)
        i = ent[1]
        ent[1] = (i + 1) & 255
    slot = ent[0][i, :n]
    slot.copy_(torch.tensor(values, dtype=torch.int32))
    return slot.to(device, non_blocking=True)
