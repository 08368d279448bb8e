"""The gradient of a forked pyramid level ACCUMULATES through the consumers that know how (round 6, layers/fused.py::_ForkAcc): a
ROIAlign backward adds its atomics to what is there, a convolution's data gradient takes what is there as its residual operand and
leaves the statistics / planes of the sum -- instead of one tensor per consumer, a zero fill per ROIAlign and a sum launch.
Whatever order autograd runs the consumers in, and with consumers that do not know the protocol among them, the result is the sum:
compared with the same graph on the summing path (MMT_FORK_ACC off, itself checked against the oracle by the step tests) and with
plain autograd in fp64 for the convolutions."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))


@pytest.fixture()
def env():
    from maskrcnn_benchmark import _hip as H
    from maskrcnn_benchmark.layers import fused
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(3)
    H.set_f16x2(True)
    keep = fused._FORK_ACC
    yield H, fused
    fused._FORK_ACC = keep
    H.set_f16x2(None)
    H.set_conv_precision(prev)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _graph(H, fused, order, x0, w3, w1, rois, levels, plain):
    """three (four with `plain`) consumers of fork(x): a 3x3 convolution (the RPN head's form), ROIAlign on two levels' worth of boxes
    (one level here), a 1x1 convolution (the hint adaptor's form), and optionally a consumer that is plain tensor arithmetic.
    `order`: the order in which the consumers are BUILT -- autograd runs their backward nodes in the reverse of it"""
    x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
    pre = fused.conv(x, torch.eye(64, device="cuda").view(64, 64, 1, 1).contiguous(memory_format=torch.channels_last))   # (makes x's aliases non-leaf)
    outs = list(fused.fork(pre, len(order) + (1 if plain else 0), ("gP", 0)))
    loss = 0.0
    for name in order:
        a = outs.pop(0)
        if name == "c3":
            loss = loss + (fused.conv(a, w3, None, 1, 1, True) * 0.5).sum()
        elif name == "c1":
            loss = loss + (fused.conv(a, w1, None, 1, 0) ** 2).sum() * 0.1
        else:
            pooled = fused.RoiAlignFpnFn.apply(rois, levels, 7, (0.25,), 2, a)
            loss = loss + (pooled * pooled).sum() * 0.01
    if plain:
        loss = loss + (outs.pop(0) * 3.0).sum()
    loss.backward()
    torch.cuda.synchronize()
    return x.grad


@pytest.mark.parametrize("order", [("c3", "roi", "c1"), ("roi", "c3"), ("roi", "roi", "c3"), ("c1", "roi"), ("roi", "c1", "roi"), ("roi", "roi")])
@pytest.mark.parametrize("plain", [False, True])
def test_accumulated_gradient_equals_the_summed_one(env, order, plain):
    H, fused = env
    g = torch.Generator().manual_seed(17)
    x0 = _cl(torch.randn(2, 64, 48, 64, generator=g).cuda())
    w3 = _cl((torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()).requires_grad_(True)
    w1 = _cl((torch.randn(64, 64, 1, 1, generator=g) * 0.1).cuda()).requires_grad_(True)
    K = 96
    xy = torch.rand(K, 2, generator=g) * torch.tensor([180.0, 120.0])
    wh = torch.rand(K, 2, generator=g) * 60 + 4
    rois = torch.cat([torch.randint(0, 2, (K, 1), generator=g).float(), xy, xy + wh], 1).cuda()
    levels = torch.zeros(K, dtype=torch.int32, device="cuda")
    sums = [0]
    real_sum = H.sum_stats

    def counting(ts, rb_site=None):
        sums[0] += 1
        return real_sum(ts, rb_site)

    H.sum_stats = counting
    try:
        fused._FORK_ACC = True
        ga = _graph(H, fused, order, x0, w3, w1, rois, levels, plain)
        n_acc, sums[0] = sums[0], 0
        w3.grad = w1.grad = None
        fused._FORK_ACC = False
        gs = _graph(H, fused, order, x0, w3, w1, rois, levels, plain)
        n_sum = sums[0]
    finally:
        H.sum_stats = real_sum
    scale = gs.abs().max().item()
    assert (ga - gs).abs().max().item() <= 2e-6 * scale, order       # (another order of fp32 additions, fp32 atomics)
    assert n_sum == 1 and n_acc == (1 if plain else 0), (order, n_acc, n_sum)   # the sum launch is gone when every consumer accumulates


def test_accumulating_consumers_leave_valid_statistics(env):
    """a convolution writes the accumulator last: what it recorded about the tensor (maximum, planes) describes the SUM; a ROIAlign adds
    to it afterwards: the records are dropped (the consumer downstream measures the tensor itself) -- never a stale record"""
    H, fused = env
    fused._FORK_ACC = True
    g = torch.Generator().manual_seed(3)
    h = fused._ForkAcc(None)
    a = _cl(torch.randn(2, 64, 16, 16, generator=g).cuda())
    a._mmt_amax = (H._amax_slot(a.device), a._version)
    assert h.put(a, True) and h.fresh and a._mmt_amax is not None
    assert not h.put(a, False) and not h.fresh and a._mmt_amax is None
    b = _cl(torch.randn(2, 64, 16, 16, generator=g).cuda())
    assert not h.put(b, True) and h.buf is b and len(h.seen) == 2
