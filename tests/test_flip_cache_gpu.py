"""_hip.weight_flip_transpose keeps the flipped fp32 weights of a Parameter for the launches between two parameter updates (the RPN
predictors' data gradient runs once per pyramid level with the same weights).  The entry must not survive an update -- also one made
through raw pointers by the library's own optimiser kernel, which bumps no tensor version -- nor another tensor at the same address."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))


def _ref(w):
    return w.detach().flip(2, 3).transpose(0, 1)


def test_flipped_weights_cached_per_parameter_generation():
    from maskrcnn_benchmark import _hip as H
    H.lib()
    g = torch.Generator().manual_seed(5)
    w = torch.nn.Parameter(torch.randn(15, 256, 1, 1, generator=g).cuda().contiguous(memory_format=torch.channels_last))
    c0 = H.C_CALLS[0]
    a = H.weight_flip_transpose(w.detach(), owner=w)
    b = H.weight_flip_transpose(w.detach(), owner=w)     # (a backward pass hands over a new Python object for the saved weight)
    assert H.C_CALLS[0] - c0 == 1 and b is a
    assert torch.equal(a, _ref(w))
    # an update through the library's SGD kernel (raw pointers: w._version does not move)
    grad, buf = torch.ones_like(w), torch.zeros_like(w)
    v0 = w._version
    H.sgd_momentum(w.detach(), grad, buf, 0.5, 0.0, 0.9, True)
    assert w._version == v0
    c = H.weight_flip_transpose(w.detach(), owner=w)
    assert torch.equal(c, _ref(w)) and not torch.equal(c, a)
    # an in-place tensor update (version moves)
    with torch.no_grad():
        w.mul_(2.0)
    d = H.weight_flip_transpose(w.detach(), owner=w)
    assert torch.equal(d, _ref(w))
    # no owner: never cached; a computed tensor (the RPN head's concatenated predictor weights) as its own owner: cached while it lives
    t = torch.randn(15, 256, 1, 1, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    c0 = H.C_CALLS[0]
    H.weight_flip_transpose(t)
    H.weight_flip_transpose(t)
    assert H.C_CALLS[0] - c0 == 2
    H.weight_flip_transpose(t.detach(), owner=t)
    H.weight_flip_transpose(t.detach(), owner=t)
    assert H.C_CALLS[0] - c0 == 3
    # another Parameter that lands on the address of a dead one
    ptr = w.data_ptr()
    shape = tuple(w.shape)
    del w, a, b, c, d
    torch.cuda.synchronize()
    w2 = torch.nn.Parameter(torch.randn(*shape, generator=g).cuda().contiguous(memory_format=torch.channels_last))
    e = H.weight_flip_transpose(w2.detach(), owner=w2)
    assert torch.equal(e, _ref(w2)), "stale entry served for a new tensor%s" % (" at the same address" if w2.data_ptr() == ptr else "")
