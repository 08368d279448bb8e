"""Round 6: the weight gradients of a BATCH of layers as grouped launches (include/mmtpsm.h: mmt_conv_wgrad_group;
csrc/conv_wgpl.hip: wgrad_pl_group_kernel, wgrad_reduce_group_kernel; csrc/conv_wgrad.hip: conv_wgrad_pipe_group_kernel).

A backward pass hands its weight-gradient jobs to the side stream a batch at a time; one by one, a layer at N = 2 is cut into as many
pixel ranges as it takes to fill the chip alone.  In a group the tiles of all layers fill it together.  Checked here, on batches like a
backward pass's (3x3 layers with both operands' row-blocked planes, 1x1 layers without, a strided 1x1, an fc layer, a 15-channel
predictor the groups do not take):
  * every dW (accumulated into a non-zero buffer, row scale applied) and bias gradient against fp64: the default arithmetic's bound;
  * against the single launches (MMT_WGRAD_GROUP=0): equal to rounding (other pixel ranges = another summation order);
  * repeatable bit for bit; more jobs than one group holds (12 plane-fed / 6 register-splitting per launch);
  * an operand whose range defeats fp16 inside a group: that job's blocks take the exact path, the others are untouched."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
sys.path.insert(0, ROOT)


@pytest.fixture()
def hip():
    from maskrcnn_benchmark import _hip as H
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(3)
    H.set_f16x2(True)
    yield H
    H.WGRAD_GROUP = True
    os.environ.pop("MMT_WGRAD_GROUP", None)
    H.set_f16x2(None)
    H.set_conv_precision(prev)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


# (N, Cin, H, W, Cout, k, stride, planes?, bias?, rowscale?)
BATCH = [
    (2, 256, 64, 64, 256, 3, 1, True, False, True),     # layer3 conv2: plane-fed
    (2, 128, 128, 128, 128, 3, 1, True, False, True),   # layer2 conv2
    (2, 512, 32, 32, 512, 3, 1, True, False, True),     # layer4 conv2
    (2, 256, 64, 64, 256, 3, 1, True, True, False),     # an FPN output conv (bias)
    (2, 1024, 64, 64, 256, 1, 1, False, False, True),   # layer3 conv1: register-splitting
    (2, 256, 64, 64, 1024, 1, 1, False, False, True),   # layer3 conv3
    (2, 512, 128, 128, 128, 1, 1, False, False, True),  # layer2 conv1
    (2, 512, 64, 64, 256, 1, 2, False, False, True),    # layer3.0 conv1 (stride 2)
    (2, 1024, 64, 64, 256, 1, 1, False, True, False),   # an FPN lateral (bias)
    (2, 256, 32, 32, 15, 1, 1, False, True, False),     # RPN predictors: 15 channels, no group takes it
    (64, 1024, 1, 1, 1024, 1, 1, False, True, False),   # fc7-like (H = W = 1: pixel-decode form 0)
]


def _make(H, spec, seed):
    N, Cin, Hh, W, Cout, k, stride, planes, bias, rsc = spec
    g = torch.Generator().manual_seed(seed)
    x = _cl(torch.randn(N, Cin, Hh, W, generator=g).relu().cuda())
    Ho, Wo = (Hh + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = _cl((torch.randn(N, Cout, Ho, Wo, generator=g) * 1e-3).cuda())
    for t in (x, dy):
        t._mmt_amax = H._amax_of(t)
        if planes:
            H.f16_split_pg(t)
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda() if rsc else None
    dw0 = _cl((torch.randn(Cout, Cin, k, k, generator=g) * 1e-5).cuda())
    db0 = torch.zeros(Cout, device="cuda") if bias else None
    return x, dy, (Cout, Cin, k, k), stride, k // 2, dw0, rs, db0


def _reference(x, dy, shape, stride, pad, rs):
    Cout, Cin, k, _ = shape
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), shape, dy.double().cpu(), stride=stride, padding=pad)
    bound = torch.nn.grad.conv2d_weight(x.double().cpu().abs(), shape, dy.double().cpu().abs(), stride=stride, padding=pad)
    if rs is not None:
        ref = ref * rs.double().cpu().view(-1, 1, 1, 1)
        bound = bound * rs.double().cpu().view(-1, 1, 1, 1)
    return ref, bound


def _run(H, specs, group, seeds=None):
    H.WGRAD_GROUP = group
    jobs, outs = [], []
    for i, sp in enumerate(specs):
        x, dy, shape, stride, pad, dw0, rs, db0 = _make(H, sp, 100 + (seeds[i] if seeds else i))
        dw = dw0.clone(memory_format=torch.preserve_format)
        db = db0.clone() if db0 is not None else None
        jobs.append((x, dy, shape, stride, pad, dw, rs, db))
        outs.append((x, dy, shape, stride, pad, dw0, rs, dw, db))
    n0 = H.F16_STATS.get("wgrad_grouped", 0)
    H.conv_wgrad_group(jobs)
    torch.cuda.synchronize()
    return outs, H.F16_STATS.get("wgrad_grouped", 0) - n0


def test_grouped_weight_gradients_of_a_backward_batch(hip):
    H = hip
    outs, n_grouped = _run(H, BATCH, True)
    assert n_grouped == len(BATCH) - 1          # everything but the 15-channel predictor went through mmt_conv_wgrad_group
    singles, n0 = _run(H, BATCH, False)
    assert n0 == 0
    again, _ = _run(H, BATCH, True)
    for sp, (x, dy, shape, stride, pad, dw0, rs, dw, db), s1, a1 in zip(BATCH, outs, singles, again):
        ref, bound = _reference(x, dy, shape, stride, pad, rs)
        got = dw.double().cpu() - dw0.double().cpu()
        err = ((got - ref).abs() / bound.clamp_min(1e-300)).max().item()
        assert err <= 3e-6, (sp, err)
        one = s1[7].double().cpu() - dw0.double().cpu()
        assert ((got - one).abs() / bound.clamp_min(1e-300)).max().item() <= 3e-6, sp
        assert torch.equal(dw, a1[7]), sp                    # repeatable bit for bit
        if db is not None:
            rb = dy.double().sum((0, 2, 3)).cpu()
            bs = dy.double().abs().sum((0, 2, 3)).max().item()
            assert (db.double().cpu() - rb).abs().max().item() <= 1e-5 * bs, sp
            assert (s1[8].double().cpu() - rb).abs().max().item() <= 1e-5 * bs, sp


def test_more_jobs_than_one_group_holds(hip):
    H = hip
    specs = [(2, 256, 32, 32, 256, 3, 1, True, False, True)] * 14 + [(2, 512, 32, 32, 128, 1, 1, False, False, True)] * 8
    outs, n_grouped = _run(H, specs, True, seeds=list(range(len(specs))))
    assert n_grouped == len(specs)
    for sp, (x, dy, shape, stride, pad, dw0, rs, dw, db) in list(zip(specs, outs))[::3]:
        ref, bound = _reference(x, dy, shape, stride, pad, rs)
        got = dw.double().cpu() - dw0.double().cpu()
        assert ((got - ref).abs() / bound.clamp_min(1e-300)).max().item() <= 3e-6, sp


def test_a_weight_shared_by_several_jobs_of_a_batch(hip):
    """the RPN head's 3x3 over the pyramid levels: jobs with ONE dw in one batch -- the items of a group run concurrently and accumulate
    without atomics, so only the first may ride in a group; the sum must be the sum"""
    H = hip
    H.WGRAD_GROUP = True
    g = torch.Generator().manual_seed(77)
    dw = _cl(torch.zeros(256, 256, 3, 3, device="cuda"))
    db = torch.zeros(256, device="cuda")
    jobs, refs = [], None
    others = [(2, 128, 64, 64, 128, 3, 1, True, False, True), (2, 512, 32, 32, 512, 3, 1, True, False, True)]
    for i, S in enumerate((128, 64, 32, 64)):
        x = _cl(torch.randn(2, 256, S, S, generator=g).relu().cuda())
        dy = _cl((torch.randn(2, 256, S, S, generator=g) * 1e-3).cuda())
        for t in (x, dy):
            t._mmt_amax = H._amax_of(t)
            H.f16_split_pg(t)
        jobs.append((x, dy, (256, 256, 3, 3), 1, 1, dw, None, db))
        r, b = _reference(x, dy, (256, 256, 3, 3), 1, 1, None)
        refs = (r, b, dy.double().sum((0, 2, 3)).cpu()) if refs is None else (refs[0] + r, refs[1] + b, refs[2] + dy.double().sum((0, 2, 3)).cpu())
        if i < len(others):   # other layers' jobs in between, as in a backward pass
            xo, dyo, sh, st, pd, dwo, rso, _ = _make(H, others[i], 500 + i)
            jobs.append((xo, dyo, sh, st, pd, dwo.clone(memory_format=torch.preserve_format), rso, None))
    H.conv_wgrad_group(jobs)
    torch.cuda.synchronize()
    assert ((dw.double().cpu() - refs[0]).abs() / refs[1].clamp_min(1e-300)).max().item() <= 3e-6
    assert (db.double().cpu() - refs[2]).abs().max().item() <= 1e-5 * refs[2].abs().max().item() + 1e-6


def test_range_guard_inside_a_group(hip):
    """one job's x has a single element 10^9 x the rest (fp16 cannot hold the tensor): its blocks take the exact fp32 path -- the planes
    of that job are useless and unused --, its neighbours in the group are what they are without it"""
    H = hip
    specs = [(2, 256, 64, 64, 256, 3, 1, True, False, True), (2, 256, 64, 64, 256, 3, 1, True, False, True),
             (2, 1024, 32, 32, 256, 1, 1, False, False, True), (2, 1024, 32, 32, 256, 1, 1, False, False, True)]
    H.WGRAD_GROUP = True
    jobs, keepers = [], []
    for i, sp in enumerate(specs):
        x, dy, shape, stride, pad, dw0, rs, db0 = _make(H, sp, 300 + i)
        if i in (1, 3):
            x = x.clone(memory_format=torch.preserve_format)
            x[-1, -1, -1, -1] = 1.0e9   # (outside the statistics' sampled blocks: inside, a lone outlier dominates the sample's own mean)
            x._mmt_amax = H._amax_of(x)
            if sp[7]:
                H.f16_split_pg(x)
        dw = dw0.clone(memory_format=torch.preserve_format)
        jobs.append((x, dy, shape, stride, pad, dw, rs, None))
        keepers.append((x, dy, shape, stride, pad, dw0, rs, dw))
    H.conv_wgrad_group(jobs)
    torch.cuda.synchronize()
    for sp, (x, dy, shape, stride, pad, dw0, rs, dw) in zip(specs, keepers):
        ref, bound = _reference(x, dy, shape, stride, pad, rs)
        got = dw.double().cpu() - dw0.double().cpu()
        assert ((got - ref).abs() / bound.clamp_min(1e-300)).max().item() <= 3e-6, sp
