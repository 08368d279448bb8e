"""Round 6: row-blocked fp16 planes out of the PRODUCERS' epilogues (include/mmtpsm.h: mmt_conv_args.y_rb, mmt_sum_stats_rb,
mmt_rb_scales_update; maskrcnn_benchmark/_hip.py: _rb_produce / f16_split_pg / rb_scales_update).

A plane-fed consumer (tap-strip kernel, plane-fed GEMM, plane-fed weight gradient) used to run a split pass over its input; the launch
that produces the tensor now writes the planes itself, with a scale fixed BEFORE the values exist (the site's maximum of the previous
step x 2).  Checked here:
  * the planes every epilogue form writes -- register-direct (tiled, tap-strip, plane-fed incl. K groups and in-launch split-K),
    LDS-staged (top-down add), split-K finish launch, the sum launch of a multi-consumer gradient -- are BIT-IDENTICAL to the split of
    the stored fp32 result with the same scale, in the row-blocked order, ragged shapes included; y itself is unchanged;
  * a site's first call (no scale yet) writes none and the consumer falls back to its split pass;
  * a consumer fed from such planes gives the result of the split-pass route (<= 3e-6 of sum |a||b| against fp64; the scale differs
    by a power of two, products are exact);
  * the range guard with the scale actually applied: a tensor that outgrew the head-room, or shrank far below it, is computed with
    exact fp32 products (still <= 3e-6) -- and the planes the slow path of a PRODUCER writes are the promised ones;
  * the plane-fed weight gradient takes both kinds of planes;
  * a whole training step with the mechanism on equals the step with it off (losses 1e-6, update 1e-5)."""
import json
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
    H.set_f16x2(True)      # (also forgets every producing site)
    yield H
    H.RB_EPI = True
    H.set_f16x2(None)
    H.set_conv_precision(prev)
    for k in ("MMT_SPLITK", "MMT_STRIP", "MMT_PG", "MMT_DIRECT_EPI"):
        os.environ.pop(k, None)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _ref_planes(y, s):
    """the two fp16 planes of y * s in the row-blocked order [N H][C / 16][W][16], as the split pass defines them"""
    N, C, Hh, W = y.shape
    r = (y.permute(0, 2, 3, 1).float() * s).clamp(-65504.0, 65504.0)          # [N][H][W][C]
    h = r.half()
    l = (r - h.float()).half()

    def rb(t):
        return t.reshape(N * Hh, W, C // 16, 16).permute(0, 2, 1, 3).contiguous().reshape(-1)
    return rb(h), rb(l)


def _site_scale(H, y):
    rb = y._mmt_rb
    assert rb[3] == "epi" and rb[2] == y._version
    return float(rb[1][0])


# (N, Cin, H, W, Cout, k, stride, pad, options, environment) -- which epilogue form each one lands in is noted; the assertions do not
# depend on it (a dispatcher change moves a case, not the contract)
PRODUCERS = [
    ((2, 512, 32, 32, 128, 1, 1, 0, ("bn", "relu"), {}), "tiled, register-direct"),
    ((8, 512, 64, 64, 128, 1, 1, 0, ("bn", "relu"), {}), "tiled 128 x 128 / 128 x 64, register-direct"),
    ((2, 256, 64, 64, 128, 1, 2, 0, ("bn", "relu"), {}), "strided 1x1 (layer2.0 conv1)"),
    ((2, 2048, 32, 32, 512, 1, 1, 0, ("mask",), {}), "tiled split-K + finish launch, masked"),
    ((2, 512, 64, 64, 256, 1, 1, 0, ("bias", "up"), {}), "LDS-staged: top-down add (res_mode 2)"),
    ((2, 1024, 30, 34, 256, 1, 1, 0, ("bias",), {}), "ragged map, tiled"),
    ((2, 256, 128, 128, 256, 3, 1, 1, ("bias",), {}), "tap-strip, register-direct"),
    ((2, 256, 128, 128, 256, 3, 1, 1, ("bias", "res"), {"MMT_DIRECT_EPI": "0"}), "tap-strip, staged epilogue"),
    ((2, 256, 64, 64, 256, 3, 1, 1, ("bias", "relu"), {}), "plane-fed, 64-row tiles (K groups)"),
    ((2, 256, 32, 32, 256, 3, 1, 1, ("bias",), {}), "plane-fed, in-launch split-K"),
    ((25, 256, 14, 14, 256, 3, 1, 1, ("bias", "relu"), {}), "plane-fed, 14 x 14 maps, ragged M"),
    ((400, 256, 14, 14, 256, 3, 1, 1, ("bias", "relu", "mask"), {}), "plane-fed, 256-row tiles, masked"),
    ((2, 256, 64, 64, 256, 3, 1, 1, ("bias",), {"MMT_PG": "0"}), "small-map 3x3 on the tiled kernel (split-K)"),
    ((2, 256, 128, 128, 256, 1, 1, 0, ("bias", "up"), {}), "row-resident K = 256, top-down add (the C2 lateral)"),
    ((2, 64, 128, 128, 256, 1, 1, 0, ("bn", "relu", "res"), {}), "row-resident K = 64, residual + ReLU"),
    ((2, 128, 64, 64, 512, 1, 1, 0, ("mask",), {}), "row-resident K = 128, masked"),
    ((3, 256, 36, 44, 128, 1, 1, 0, ("bias",), {}), "row-resident, ragged rows (M % 128 != 0)"),
]


def _make(case, seed=0):
    N, C, Hh, W, Co, k, stride, pad, opts, _env = case
    g = torch.Generator().manual_seed(seed + N + C + Hh + Co + k)
    x = _cl(torch.randn(N, C, Hh, W, generator=g).relu().cuda())
    w = _cl((torch.randn(Co, C, k, k, generator=g) * (2.0 / (k * k * C)) ** 0.5).cuda())
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    sc = (torch.rand(Co, generator=g) + 0.5).cuda() if "bn" in opts else None
    sh = (torch.randn(Co, generator=g) * 0.1).cuda() if ("bn" in opts or "bias" in opts) else None
    kw = dict(relu="relu" in opts)
    if "res" in opts:
        kw.update(res=_cl(torch.randn(N, Co, Ho, Wo, generator=g).cuda()), res_mode=1)
    if "up" in opts:
        kw.update(res=_cl(torch.randn(N, Co, Ho // 2, Wo // 2, generator=g).cuda()), res_mode=2)
    if "mask" in opts:
        kw.update(mask=_cl(torch.randn(N, Co, Ho, Wo, generator=g).cuda()), mask_scale=2.0)
    return x, w, sc, sh, stride, pad, kw


@pytest.mark.parametrize("case,what", PRODUCERS, ids=[w for _, w in PRODUCERS])
def test_epilogue_planes_equal_the_split_of_the_result(hip, case, what):
    H = hip
    for k, v in case[9].items():
        os.environ[k] = v
    try:
        x, w, sc, sh, stride, pad, kw = _make(case)
        site = ("t", case[:8])
        y0 = H.conv_forward(x, w, sc, sh, stride, pad, rb_site=site, **kw)
        assert getattr(y0, "_mmt_rb", None) is None          # a site's first call: no scale yet, no planes
        H.rb_scales_update()
        y1 = H.conv_forward(x, w, sc, sh, stride, pad, rb_site=site, **kw)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)                           # the plane store changes nothing else
        rb = getattr(y1, "_mmt_rb", None)
        assert rb is not None, "the launch of this shape wrote no planes: " + what
        s = _site_scale(H, y1)
        amax = float(y1.abs().max())
        assert 2.0 ** 12 <= amax * s < 2.0 ** 13              # 2 x head-room below [2^13, 2^14)
        h, l = _ref_planes(y1, s)
        assert torch.equal(rb[0][0].view(torch.int16), h.view(torch.int16)), what
        assert torch.equal(rb[0][1].view(torch.int16), l.view(torch.int16)), what
        # the same with the split pass's own scale: the planes a consumer would have made itself (another power of two, same values)
        xp, st, layout, lag = H.f16_split_pg(y0)
        assert lag == 0 and layout == 1
        torch.cuda.synchronize()
        back_e = rb[0][0].float() / s + rb[0][1].float() / s
        back_s = xp[0].float() / float(st[0]) + xp[1].float() / float(st[0])
        assert (back_e - back_s).abs().max().item() <= 2.0 ** -24 * amax
    finally:
        for k in case[9]:
            os.environ.pop(k, None)


def test_planes_for_the_leading_images_only(hip):
    """the teacher's K x flip batch: only view 0 (the first n images) feeds a plane-fed launch -- the producer writes planes for those
    images' pixels alone; a batch slice inside them carries the planes, the whole tensor does not"""
    H = hip
    from maskrcnn_benchmark.layers import fused
    g = torch.Generator().manual_seed(21)
    for shape, k in (((8, 256, 64, 64), 3), ((8, 256, 128, 128), 3), ((8, 512, 32, 32), 1)):
        x = _cl(torch.randn(*shape, generator=g).cuda())
        w = _cl((torch.randn(256, shape[1], k, k, generator=g) * 0.05).cuda())
        b = torch.randn(256, generator=g).cuda()
        site = ("P", shape, H.RbLead(2))
        H.conv_forward(x, w, None, b, 1, k // 2, rb_site=site)
        H.rb_scales_update()
        y = H.conv_forward(x, w, None, b, 1, k // 2, rb_site=site)
        torch.cuda.synchronize()
        rb = y._mmt_rb
        assert rb[3] == "epi" and rb[4] == 2 and rb[0].shape[1] == y[:2].numel()
        s = float(rb[1][0])
        h, l = _ref_planes(y[:2], s)
        assert torch.equal(rb[0][0].view(torch.int16), h.view(torch.int16)) and torch.equal(rb[0][1].view(torch.int16), l.view(torch.int16))
        v = fused.batch_slice(y, 0, 2)
        assert H.f16_split_pg(v)[3] == 1                      # the slice: the producer's planes
        assert H.f16_split_pg(fused.batch_slice(y, 2, 4))[3] == 0   # outside: a split pass
        v1 = fused.batch_slice(y, 1, 2)
        xp = H.f16_split_pg(v1)
        assert xp[3] == 1
        h1, l1 = _ref_planes(y[1:2], s)
        assert torch.equal(xp[0][0].view(torch.int16), h1.view(torch.int16))
        assert H.f16_split_pg(y)[3] == 0                            # the whole batch: a split pass (which then owns the tensor's planes)


def _conv64(x, w, stride, pad):
    return F.conv2d(x.double().cpu(), w.double().cpu(), None, stride, pad)


@pytest.mark.parametrize("shape", [(2, 256, 128, 128, 256), (2, 256, 32, 32, 256), (25, 256, 14, 14, 256), (2, 512, 32, 32, 512)])
def test_consumers_fed_from_a_producers_planes(hip, shape):
    """producer (1x1) -> consumer (3x3: tap-strip or plane-fed): the consumer's result from the producer's planes against the
    split-pass route and against fp64"""
    H = hip
    N, C, Hh, W, Co = shape
    g = torch.Generator().manual_seed(3)
    x0 = _cl(torch.randn(N, 512, Hh, W, generator=g).relu().cuda())
    w1 = _cl((torch.randn(C, 512, 1, 1, generator=g) * (2.0 / 512) ** 0.5).cuda())
    w2 = _cl((torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda())
    site = ("o1", 7)
    H.conv_forward(x0, w1, relu=True, rb_site=site)
    H.rb_scales_update()
    H.RB_EPI = False
    t0 = H.conv_forward(x0, w1, relu=True, rb_site=site)
    y_split = H.conv_forward(t0, w2, None, None, 1, 1)
    H.RB_EPI = True
    n_epi = H.F16_STATS.get("rb_epi", 0)
    t1 = H.conv_forward(x0, w1, relu=True, rb_site=site)
    assert t1._mmt_rb[3] == "epi"
    y_epi = H.conv_forward(t1, w2, None, None, 1, 1)
    assert H.F16_STATS.get("rb_epi", 0) == n_epi + 1          # no split pass in front of the 3x3 launch
    torch.cuda.synchronize()
    ref = _conv64(t1, w2, 1, 1)
    bound = _conv64(t1.abs(), w2.abs(), 1, 1)
    for y in (y_split, y_epi):
        assert ((y.double().cpu() - ref).abs() / bound.clamp_min(1e-30)).max().item() <= 3e-6
    assert (y_epi - y_split).abs().max().item() <= 1e-6 * float(y_split.abs().max())


@pytest.mark.parametrize("how", ["outgrown", "shrunk"])
@pytest.mark.parametrize("shape", [(2, 256, 128, 128), (2, 256, 32, 32)])
def test_guard_with_the_scale_actually_applied(hip, how, shape):
    """the site's scale comes from the previous step: a tensor 10^4 x larger saturates the planes, one 2^-24 x smaller falls below the
    low term's range -- the consumer sees both in the producer's statistics and computes with exact fp32 products"""
    H = hip
    N, C, Hh, W = shape
    g = torch.Generator().manual_seed(5)
    x0 = _cl(torch.randn(N, 512, Hh, W, generator=g).relu().cuda())
    w1 = _cl((torch.randn(C, 512, 1, 1, generator=g) * (2.0 / 512) ** 0.5).cuda())
    w2 = _cl((torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda())
    site = ("o1", how, shape)
    H.conv_forward(x0, w1, relu=True, rb_site=site)
    H.rb_scales_update()
    x1 = _cl(x0 * (1.0e4 if how == "outgrown" else 2.0 ** -24))
    t = H.conv_forward(x1, w1, relu=True, rb_site=site)
    assert t._mmt_rb[3] == "epi"
    y = H.conv_forward(t, w2, None, None, 1, 1)
    torch.cuda.synchronize()
    ref = _conv64(t, w2, 1, 1)
    bound = _conv64(t.abs(), w2.abs(), 1, 1)
    assert ((y.double().cpu() - ref).abs() / bound.clamp_min(1e-300)).max().item() <= 3e-6, how
    # the next step's scale follows the tensor
    H.rb_scales_update()
    t2 = H.conv_forward(x1, w1, relu=True, rb_site=site)
    torch.cuda.synchronize()
    s = _site_scale(H, t2)
    assert 2.0 ** 12 <= float(t2.abs().max()) * s < 2.0 ** 13


def test_producer_on_its_slow_path_writes_the_promised_planes(hip):
    """a producer whose OWN input defeats fp16 (one element 10^9 x the rest) runs its exact fp32 path -- and still leaves the planes"""
    H = hip
    g = torch.Generator().manual_seed(9)
    x0 = _cl(torch.randn(2, 512, 32, 32, generator=g).relu().cuda())
    w1 = _cl((torch.randn(128, 512, 1, 1, generator=g) * (2.0 / 512) ** 0.5).cuda())
    site = ("slow", 1)
    x0[0, 0, 0, 0] = 1.0e9
    H.conv_forward(x0, w1, relu=True, rb_site=site)
    H.rb_scales_update()
    y = H.conv_forward(x0, w1, relu=True, rb_site=site)
    torch.cuda.synchronize()
    s = _site_scale(H, y)
    h, l = _ref_planes(y, s)
    assert torch.equal(y._mmt_rb[0][0].view(torch.int16), h.view(torch.int16))
    assert torch.equal(y._mmt_rb[0][1].view(torch.int16), l.view(torch.int16))


@pytest.mark.parametrize("shape,n", [((2, 256, 128, 128), 2), ((2, 256, 64, 64), 3), ((2, 256, 30, 50), 4), ((2, 256, 256, 256), 2)])
def test_sum_launch_leaves_the_planes_of_the_sum(hip, shape, n):
    H = hip
    g = torch.Generator().manual_seed(11)
    ts = [_cl((torch.randn(*shape, generator=g) * 10.0 ** (-3 + i)).cuda()) for i in range(n)]
    site = ("gP", shape, n)
    y0 = H.sum_stats(ts, site)
    assert getattr(y0, "_mmt_rb", None) is None
    H.rb_scales_update()
    y1 = H.sum_stats(ts, site)
    torch.cuda.synchronize()
    want = ts[0] + ts[1]
    for t in ts[2:]:
        want = want + t
    assert torch.equal(y0, want) and torch.equal(y1, want)
    s = _site_scale(H, y1)
    assert 2.0 ** 12 <= float(y1.abs().max()) * s < 2.0 ** 13
    h, l = _ref_planes(y1, s)
    assert torch.equal(y1._mmt_rb[0][0].view(torch.int16), h.view(torch.int16))
    assert torch.equal(y1._mmt_rb[0][1].view(torch.int16), l.view(torch.int16))
    # statistics as the plain sum launch records them: the maximum exactly, the sampled mean within the sampling's spread
    slot = y1._mmt_amax[0]
    row = slot.pool.dev[slot.idx]
    assert float(row[0]) == float(y1.abs().max())
    mean = float(row[1:17].sum() / row[17:33].sum())
    assert abs(mean / float(y1.abs().mean()) - 1.0) < 0.2


def test_plane_fed_weight_gradient_takes_a_producers_planes(hip):
    """o1 = relu(conv1x1(x)) with planes from the epilogue, dy with planes from a (masked) data-gradient launch: dW of the 3x3 layer
    from those planes against fp64"""
    H = hip
    if not H.WG_PLANES:
        pytest.skip("plane-fed weight gradient switched off")
    g = torch.Generator().manual_seed(13)
    N, C, Hh, W = 2, 256, 64, 64
    x0 = _cl(torch.randn(N, 1024, Hh, W, generator=g).relu().cuda())
    w1 = _cl((torch.randn(C, 1024, 1, 1, generator=g) * (2.0 / 1024) ** 0.5).cuda())
    w3 = _cl((torch.randn(1024, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).cuda())
    g3 = _cl((torch.randn(N, 1024, Hh, W, generator=g) * 1e-3).cuda())
    from maskrcnn_benchmark.layers import fused
    for rnd in range(2):
        o1 = H.conv_forward(x0, w1, relu=True, rb_site=("o1", 21))
        o2 = _cl(torch.randn(N, C, Hh, W, generator=torch.Generator().manual_seed(17)).cuda())
        d_o2 = fused._dgrad(g3, w3, o2.shape, 1, 0, None, mask=o2, rb_site=("d_o2", 21))
        H.rb_scales_update()
    assert o1._mmt_rb[3] == "epi" and d_o2._mmt_rb[3] == "epi"
    dw = _cl(torch.zeros(C, C, 3, 3, device="cuda"))
    n0 = H.F16_STATS.get("wgrad_pl", 0)
    H.conv_wgrad(o1, d_o2, (C, C, 3, 3), 1, 1, dw)
    assert H.F16_STATS.get("wgrad_pl", 0) == n0 + 1
    torch.cuda.synchronize()
    xd, gd = o1.double().cpu(), d_o2.double().cpu()
    ref = torch.nn.grad.conv2d_weight(xd, (C, C, 3, 3), gd, stride=1, padding=1)
    bound = torch.nn.grad.conv2d_weight(xd.abs(), (C, C, 3, 3), gd.abs(), stride=1, padding=1)
    assert ((dw.double().cpu() - ref).abs() / bound.clamp_min(1e-300)).max().item() <= 3e-6


def test_training_steps_with_and_without_epilogue_planes_agree():
    """three iterations of the real trainer (mean-teacher phase) at 320^2: the mechanism on (planes from the second step on) against
    off.  Planes with another power-of-two scale differ from the split pass's in the last bits of the smallest elements, and an
    un-replayed run amplifies last-bit differences through its discrete decisions (NMS ties, sampled sets): step 1 -- no planes yet --
    is identical (to the order of the loss sums' atomics), steps 2-3 agree to a few per cent.  The STRICT statement -- a step with epilogue planes against the oracle at
    1e-4 with the decisions replayed -- is tests/test_train_step_gpu.py::test_full_step_matches_oracle[...epilogue-planes]; this test
    catches a mechanism that is grossly wrong in the un-replayed trainer."""
    import synthetic
    from maskrcnn_benchmark import _hip as H
    import bench
    outs = []
    for on in (False, True):
        H.set_f16x2(True)
        H.RB_EPI = on
        try:
            torch.manual_seed(0)
            cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, crop=320, n_inst=4, base_lr=1e-4)
            losses = []
            for i in range(3):
                il, tg, ul = batch()
                losses.append({k: float(v) for k, v in trainer.train_step(1400 + i, il, tg, ul).items()})
            torch.cuda.synchronize()
            outs.append((losses, trainer.flat_s.data.clone(), trainer.flat_t.data.clone(), dict(H.F16_STATS)))
        finally:
            H.RB_EPI = True
    (l0, s0, t0, st0), (l1, s1, t1, st1) = outs
    assert st1.get("rb_epi", 0) > 0
    for k in l0[0]:                             # the first step: no site has a scale yet -- equal up to the order of the losses' atomics
        assert abs(l0[0][k] - l1[0][k]) <= 2e-6 * max(1.0, abs(l0[0][k])), (k, l0[0][k], l1[0][k])
    for a, b in zip(l0[1:], l1[1:]):
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 5e-2 * max(0.1, abs(a[k])), (k, a[k], b[k])
    assert (s0 - s1).norm().item() <= 1e-4 * s0.norm().item()
    assert (t0 - t1).norm().item() <= 1e-4 * t0.norm().item()
