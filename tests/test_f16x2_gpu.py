"""The two-term fp16 split (default arithmetic of mode 3 since round 3; include/mmtpsm.h: mmt_conv3x3_strip_f16x2,
mmt_conv_forward_f16x2; `_hip.set_f16x2(False)` / MMT_F16X2=0 selects the 3-term bf16 split it falls back to): three matrix
products per multiply instead of six.

  * kernel level, forward and data gradient, every epilogue operand, both strip widths, the split-K form: error against an
    fp64 convolution no larger than the shipped 3-term bf16 split's in the mean (rms <= 1.1 x) and within its own bound
    at the worst element (<= 2 x, far inside the mode's tolerance), for activation-like,
    signed, gradient-like (six decades) and extreme-scale (1e-20, 1e15) operands -- the scale of each tensor is derived on
    the device from its own largest magnitude, so the magnitude of a tensor is irrelevant;
  * the weight-plane cache belongs to ONE tensor object (an address reused by the allocator must not hit);
  * the detector: supervised losses against the fp32 CPU oracle at the tolerance of the default arithmetic (1e-4), gradients
    of the parameters against the default arithmetic's."""
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
    yield H
    H.set_f16x2(None)
    H.set_conv_precision(prev)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _inputs(kind, shape, g):
    t = torch.randn(shape, generator=g)
    if kind == "act":
        t = t.relu()
    elif kind == "signed50":
        t = t * 50.0
    elif kind == "grad":
        t = t * torch.exp(torch.randn(shape, generator=g) * 2.0) * 1e-5
    elif kind == "tiny":
        t = t * 1e-20
    elif kind == "huge":
        t = t * 1e15
    return _cl(t.cuda())


CASES = [  # N, Cin, H, W, Cout, input kind, epilogue
    (2, 128, 128, 128, 192, "act", "res"), (8, 256, 64, 64, 256, "signed50", "relu"), (32, 128, 32, 64, 128, "act", "mask"),
    (2, 256, 64, 64, 256, "grad", ""), (2, 128, 128, 128, 128, "grad", "mask"), (2, 256, 64, 64, 256, "tiny", ""),
    (2, 256, 128, 128, 256, "huge", "relu"), (2, 256, 256, 256, 256, "act", "relu"),
]


@pytest.mark.parametrize("case", CASES)
def test_strip_f16x2_error_not_above_bf16x3(hip, case):
    H = hip
    N, C, Hh, W, Co, kind, opts = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = _inputs(kind, (N, C, Hh, W), g)
    w = _cl((torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda())
    mag = x.abs().max().item() * 0.3
    sc, sh = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.randn(Co, generator=g) * 0.1 * mag).cuda()
    res = _cl((torch.randn(N, Co, Hh, W, generator=g) * mag).cuda()) if opts == "res" else None
    mask = _cl(torch.randn(N, Co, Hh, W, generator=g).cuda()) if opts == "mask" else None
    kw = dict(relu=opts in ("res", "relu"), res=res, res_mode=1 if res is not None else 0, mask=mask, mask_scale=2.0)
    H.set_f16x2(False)
    y3 = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
    H.set_f16x2(True)
    yh = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
    H.set_f16x2(False)
    ref = F.conv2d(x[:1].double(), w.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res[:1].double()
    if kw["relu"]:
        ref = F.relu(ref)
    if mask is not None:
        ref = torch.where(mask[:1] > 0, ref * 2.0, torch.zeros_like(ref))
    scale = ref.abs().max().item()
    e3 = (y3[:1].double() - ref).abs().max().item() / scale
    eh = (yh[:1].double() - ref).abs().max().item() / scale
    r3 = (y3[:1].double() - ref).pow(2).mean().sqrt().item() / scale
    rh = (yh[:1].double() - ref).pow(2).mean().sqrt().item() / scale
    assert not torch.equal(y3, yh)                 # it really ran the other arithmetic
    assert eh < 1e-5, (case, eh)                   # the tolerance of the default mode (tests/test_hip_kernels.py: MODE_TOL[3])
    assert rh <= 1.1 * r3 + 1e-9, (case, rh, r3)   # no worse than it in the mean ...
    assert eh <= 2.0 * e3 + 1e-7, (case, eh, e3)   # ... nor at the worst element (the bound the strip kernel's own test uses)


@pytest.mark.parametrize("case", CASES + [(3, 256, 64, 192, 256, "act", "relu"), (2, 512, 64, 64, 128, "act", "")])
def test_strip_row_blocked_planes_same_bits(hip, case, monkeypatch):
    """the tap-strip kernel fed from row-blocked input planes [N H][Cin / 16][W][16] (round 5, the default: an item's 32 pixels are
    one contiguous run) returns the bits it returns from planes indexed like x; full-width, 64-pixel and split-K forms"""
    H = hip
    N, C, Hh, W, Co, kind, opts = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = _inputs(kind, (N, C, Hh, W), g)
    w = _cl((torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda())
    sc, sh = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.randn(Co, generator=g) * 0.1).cuda()
    res = _cl(torch.randn(N, Co, Hh, W, generator=g).cuda()) if opts == "res" else None
    mask = _cl(torch.randn(N, Co, Hh, W, generator=g).cuda()) if opts == "mask" else None
    kw = dict(relu=opts in ("res", "relu"), res=res, res_mode=1 if res is not None else 0, mask=mask, mask_scale=2.0)
    H.set_f16x2(True)
    try:
        assert H.PG_RB
        n0 = H.F16_STATS["conv"] + H.F16_STATS["pg"]
        y1 = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
        y1b = H.conv_forward(x, w, sc, sh, 1, 1, **kw)       # (second call: the cached-plan path)
        monkeypatch.setattr(H, "PG_RB", False)
        y0 = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
        assert H.F16_STATS["conv"] + H.F16_STATS["pg"] == n0 + 3     # a plane-fed kernel (tap-strip; few-tile shapes: the implicit GEMM), three times
        assert torch.equal(y0, y1) and torch.equal(y0, y1b)
    finally:
        H.set_f16x2(False)


def test_strip_f16x2_data_gradient(hip):
    """the data gradient of a 3x3 convolution (flipped, transposed, BN-scaled weights packed as fp16 terms on the device)"""
    H = hip
    from maskrcnn_benchmark.layers import fused
    g = torch.Generator().manual_seed(9)
    N, C, S, Co = 2, 256, 64, 256
    w = _cl((torch.randn(Co, C, 3, 3, generator=g) * 0.02).cuda())
    bn = (torch.rand(Co, generator=g) + 0.5).cuda()
    dy = _inputs("grad", (N, Co, S, S), g)
    act = _inputs("act", (N, C, S, S), g)
    outs = {}
    for on in (False, True):
        H.set_f16x2(on)
        outs[on] = fused._dgrad(dy, w, (N, C, S, S), 1, 1, bn, mask=act)
    H.set_f16x2(False)
    wd = (w.double() * bn.double().view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1)
    ref = F.conv2d(dy[:1].double(), wd, None, 1, 1) * (act[:1] > 0)
    scale = ref.abs().max().item()
    e3 = (outs[False][:1].double() - ref).abs().max().item() / scale
    eh = (outs[True][:1].double() - ref).abs().max().item() / scale
    assert not torch.equal(outs[False], outs[True])
    assert eh < 1e-5 and eh <= 2.0 * e3 + 1e-7, (eh, e3)


@pytest.mark.parametrize("shape", [(2, 256, 64, 64, 256, 3, "act", "grad"), (2, 128, 128, 128, 128, 3, "act", "grad"),
                                   (2, 512, 32, 32, 512, 3, "act", "act"), (1, 256, 32, 96, 128, 3, "signed50", "grad"),
                                   (3, 128, 32, 32, 256, 1, "act", "grad"), (2, 256, 64, 64, 256, 3, "huge", "tiny"),
                                   (2, 128, 64, 64, 128, 5, "act", "grad")])
def test_wgrad_from_row_blocked_planes(hip, shape, monkeypatch):
    """mmt_conv_wgrad_planes (csrc/conv_wgpl.hip): the weight gradient of a stride-1 layer from the row-blocked fp16 planes both
    operands already have -- LDS-DMA of 1 KiB runs, ds_read_b64_tr_b16 fragments, no vector arithmetic -- against fp64 within the
    default arithmetic's bound and against the register-splitting kernel (same products, another summation order); accumulation
    into a non-zero buffer, row scale, bias gradient, one and several pixel ranges across blocks, batch slices of the planes"""
    H = hip
    N, Cin, Hh, W, Cout, k, kx, kd = shape
    g = torch.Generator().manual_seed(sum(shape[:6]))
    x = _inputs(kx, (N, Cin, Hh, W), g)
    dy = _inputs(kd, (N, Cout, Hh, W), g)
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda()
    dw0 = _cl((torch.randn((Cout, Cin, k, k), generator=g) * 1e-3 * float(x.abs().max() * dy.abs().max())).cuda())
    H.set_f16x2(True)
    try:
        for t in (x, dy):
            t._mmt_amax = H._amax_of(t)
            H.f16_split_pg(t)           # what the forward / data-gradient launch of the layer leaves behind
        assert x._mmt_rb[2] == x._version and dy._mmt_rb[2] == dy._version

        def run(planes, xx=x, dd=dy):
            monkeypatch.setattr(H, "WG_PLANES", planes)
            dw = dw0.clone(memory_format=torch.preserve_format)
            db = torch.zeros((Cout,), device="cuda")
            H.conv_wgrad(xx, dd, (Cout, Cin, k, k), 1, k // 2, dw, rs, db)
            torch.cuda.synchronize()
            return dw, db
        n0 = H.F16_STATS.get("wgrad_pl", 0)
        d_pl, b_pl = run(True)
        assert H.F16_STATS.get("wgrad_pl", 0) == n0 + 1
        d_old, b_old = run(False)
        assert H.F16_STATS.get("wgrad_pl", 0) == n0 + 1
        xu = F.unfold(x.double(), k, padding=k // 2)
        ref = torch.einsum("nco,nko->ck", dy.double().flatten(2), xu).view(Cout, Cin, k, k) * rs.double().view(-1, 1, 1, 1)
        scale = ref.abs().max().item()
        e_pl = (d_pl.double() - dw0.double() - ref).abs().max().item() / scale
        e_old = (d_old.double() - dw0.double() - ref).abs().max().item() / scale
        assert e_pl < 1e-5 and e_pl <= 2.0 * e_old + 2e-7, (e_pl, e_old)
        rb = dy.double().sum((0, 2, 3))
        bs = dy.double().abs().sum((0, 2, 3)).max().item()
        assert (b_pl.double() - rb).abs().max().item() < 1e-5 * bs and (b_old.double() - rb).abs().max().item() < 1e-5 * bs
        if N > 1:   # a batch slice of the tensors (the pair schedule's backward: N = 2 views of the N = 4 forward) takes sliced planes
            from maskrcnn_benchmark.layers import fused
            xs, ds = fused.batch_slice(x, 1, N), fused.batch_slice(dy, 1, N)
            d_s, _ = run(True, xs, ds)
            assert H.F16_STATS.get("wgrad_pl", 0) == n0 + 2
            xu = F.unfold(x[1:].double(), k, padding=k // 2)
            r_s = torch.einsum("nco,nko->ck", dy[1:].double().flatten(2), xu).view(Cout, Cin, k, k) * rs.double().view(-1, 1, 1, 1)
            assert (d_s.double() - dw0.double() - r_s).abs().max().item() < 1e-5 * max(r_s.abs().max().item(), 1e-30)
    finally:
        H.set_f16x2(False)


@pytest.mark.parametrize("shape", [(2, 128, 64, 64, 128, 3, "act", "grad"), (2, 256, 64, 64, 256, 3, "act", "grad"),
                                   (2, 256, 128, 128, 256, 3, "signed50", "tiny"), (8, 128, 32, 32, 128, 3, "act", "act")])
def test_wgrad_f16x2(hip, shape):
    """weight gradient with both operands carrying their recorded maximum (`_mmt_amax`, as the producing launches attach it):
    two-term fp16 split, 3 products; error against fp64 within the default's bound, split and un-split accumulation"""
    H = hip
    N, Cin, Hh, W, Cout, k, kx, kd = shape
    g = torch.Generator().manual_seed(sum(shape[:6]))
    x = _inputs(kx, (N, Cin, Hh, W), g)
    dy = _inputs(kd, (N, Cout, Hh, W), g)
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda()

    def run(on):
        H.set_f16x2(on)
        if on:
            x._mmt_amax = (x.abs().max().reshape(1), x._version)
            dy._mmt_amax = (dy.abs().max().reshape(1), dy._version)
        dw = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
        H.conv_wgrad(x, dy, (Cout, Cin, k, k), 1, k // 2, dw, rs, None)
        H.set_f16x2(False)
        return dw
    d3, dh = run(False), run(True)
    xu = F.unfold(x.double(), k, padding=k // 2)                      # (N, Cin k k, HW)
    ref = torch.einsum("nco,nko->ck", dy.double().flatten(2), xu).view(Cout, Cin, k, k) * rs.double().view(-1, 1, 1, 1)
    scale = ref.abs().max().item()
    e3 = (d3.double() - ref).abs().max().item() / scale
    eh = (dh.double() - ref).abs().max().item() / scale
    assert not torch.equal(d3, dh)
    assert eh < 2e-5 and eh <= 2.0 * e3 + 1e-7, (shape, eh, e3)


@pytest.mark.parametrize("shape", [(2, 256, 64, 64, 256, 3), (2, 1024, 64, 64, 256, 1), (2, 128, 128, 128, 128, 3), (1, 256, 32, 32, 512, 1)])
def test_wgrad_two_segments(hip, shape):
    """mmt_conv_args.x2 / dy2 (round 4): the weight gradients of TWO passes through one layer as one launch -- every block works on
    one segment with that segment's own power-of-two scales (the operands of the second pass are 10^4 x smaller here), bias sums
    included.  Against fp64 within the fp16 split's bound, against two separate launches to fp32 rounding of the sums, and with an
    outlier in the second segment only (its blocks take the exact path, the first segment's stay fast)."""
    H = hip
    N, Cin, S, _, Cout, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    x1, x2 = _inputs("act", (N, Cin, S, S), g), _inputs("act", (N, Cin, S, S), g) * 1e-4
    d1, d2 = _inputs("grad", (N, Cout, S, S), g), _inputs("grad", (N, Cout, S, S), g) * 1e-4
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda()
    H.set_f16x2(True)

    def stats(*ts):
        for t in ts:
            t._mmt_amax = H._amax_of(t)

    def ref(xa, da, xb, db_):
        out = 0
        for xx, dd in ((xa, da), (xb, db_)):
            xu = F.unfold(xx.double(), k, padding=k // 2)
            out = out + torch.einsum("nco,nko->ck", dd.double().flatten(2), xu).view(Cout, Cin, k, k)
        return out * rs.double().view(-1, 1, 1, 1)

    stats(x1, x2, d1, d2)
    assert H.wgrad_pair_ok(x1, d1, x2, d2)
    n0 = H.F16_STATS.get("wgrad_pairs", 0)
    dw = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
    db = torch.zeros((Cout,), device="cuda")
    H.conv_wgrad(x1, d1, (Cout, Cin, k, k), 1, k // 2, dw, rs, db, pair=(x2, d2))
    assert H.F16_STATS["wgrad_pairs"] == n0 + 1
    dws = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
    dbs = torch.zeros((Cout,), device="cuda")
    H.conv_wgrad(x1, d1, (Cout, Cin, k, k), 1, k // 2, dws, rs, dbs)
    H.conv_wgrad(x2, d2, (Cout, Cin, k, k), 1, k // 2, dws, rs, dbs)
    r = ref(x1, d1, x2, d2)
    scale = r.abs().max().item()
    assert (dw.double() - r).abs().max().item() < 2e-5 * scale
    assert (dw - dws).abs().max().item() < 2e-6 * scale          # the same products, summed in another order
    rb = (d1.double() + d2.double()).sum((0, 2, 3))
    assert (db.double() - rb).abs().max().item() < 1e-5 * (d1.double().abs() + d2.double().abs()).sum((0, 2, 3)).max().item()
    # the second segment's contribution is really there (10^-8 of the first: check it alone through linearity)
    dw2 = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
    H.conv_wgrad(x2, d2, (Cout, Cin, k, k), 1, k // 2, dw2, rs, None, pair=(x2, d2))
    r2 = ref(x2, d2, x2, d2)
    assert (dw2.double() - r2).abs().max().item() < 2e-5 * r2.abs().max().item()
    # an outlier in the SECOND pass's gradient only
    d2o = d2.clone()
    d2o[N // 2, 17, S // 2 - 2, S // 2 - 1] = 1.0e8 * 1e-9
    stats(d2o)
    dwo = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
    H.conv_wgrad(x1, d1, (Cout, Cin, k, k), 1, k // 2, dwo, rs, None, pair=(x2, d2o))
    ro = ref(x1, d1, x2, d2o)
    den = 0
    for xx, dd in ((x1, d1), (x2, d2o)):
        den = den + torch.einsum("nco,nko->ck", dd.double().abs().flatten(2), F.unfold(xx.double(), k, padding=k // 2).abs()).view(Cout, Cin, k, k)
    den = den * rs.double().view(-1, 1, 1, 1) + 1e-300
    assert ((dwo.double() - ro).abs() / den).max().item() < 3e-6


ROWS_CASES = [  # N, Cin, H, W, Cout, epilogue: the row-resident 1x1 kernel on the fp16 split (K = 64 / 128 / 256)
    (8, 256, 64, 64, 1024, "res"), (2, 256, 64, 64, 1024, "res"), (2, 256, 64, 64, 1024, "mask"), (1, 256, 47, 45, 200, "relu"),
    (4, 64, 128, 128, 256, "res"), (1, 64, 257, 259, 96, ""), (2, 128, 128, 128, 512, "maskres"), (8, 256, 256, 256, 128, ""),
    (2, 256, 32, 32, 2048, "mask"), (8, 256, 64, 96, 256, "up2"), (2, 256, 256, 256, 256, "up2"),
]


@pytest.mark.parametrize("case", ROWS_CASES)
def test_rows_kernel_f16x2(hip, case):
    """conv1x1_rows_kernel<KT, 32, 2, true>: 1x1 layers with K = 64 / 128 / 256 keep their rows in registers as (h, l) fp16
    fragments and walk the output panels (column groups when the row blocks alone would not fill the chip): against fp64
    within the default mode's bound, no worse than the 3-term bf16 split, and equal to fp32 rounding to the tiled kernel on
    the same arithmetic (MMT_ROWS=0); residual, ReLU, the (x > 0) mask of a data gradient, ragged M, Cout % 32 != 0"""
    H = hip
    N, C, Hh, W, Co, opts = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = _inputs("act" if "mask" not in opts else "grad", (N, C, Hh, W), g)
    w = _cl((torch.randn(Co, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).cuda())
    mag = x.abs().max().item() * 0.3
    sc, sh = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.randn(Co, generator=g) * 0.1 * mag).cuda()
    res = _cl((torch.randn(N, Co, Hh, W, generator=g) * mag).cuda()) if "res" in opts else None
    mask = _cl(torch.randn(N, Co, Hh, W, generator=g).cuda()) if "mask" in opts else None
    kw = dict(relu=opts in ("res", "relu"), res=res, res_mode=1 if res is not None else 0, mask=mask, mask_scale=2.0)
    if opts == "up2":   # the FPN lateral: + the coarser level, nearest-x2 upsampled (res_mode 2)
        top = _cl((torch.randn(N, Co, Hh // 2, W // 2, generator=g) * mag).cuda())
        kw.update(res=top, res_mode=2)
        res = F.interpolate(top, scale_factor=2, mode="nearest")
    H.set_f16x2(False)
    y3 = H.conv_forward(x, w, sc, sh, 1, 0, **kw)
    H.set_f16x2(True)
    n0 = H.F16_STATS["tiled"]
    yh = H.conv_forward(x, w, sc, sh, 1, 0, **kw)
    assert H.F16_STATS["tiled"] == n0 + 1
    old = os.environ.get("MMT_ROWS")
    os.environ["MMT_ROWS"] = "0"
    try:
        yt = H.conv_forward(x, w, sc, sh, 1, 0, **kw)
    finally:
        if old is None:
            os.environ.pop("MMT_ROWS", None)
        else:
            os.environ["MMT_ROWS"] = old
    H.set_f16x2(False)
    ref = F.conv2d(x.double(), w.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res.double()
    if kw["relu"]:
        ref = F.relu(ref)
    if mask is not None:
        ref = torch.where(mask > 0, ref * 2.0, torch.zeros_like(ref))
    scale = ref.abs().max().item()
    e3 = (y3.double() - ref).abs().max().item() / scale
    eh = (yh.double() - ref).abs().max().item() / scale
    assert eh < 1e-5 and eh <= 2.0 * e3 + 2e-7, (case, eh, e3)
    assert not torch.equal(yh, y3)
    assert (yh - yt).abs().max().item() <= 4e-6 * scale, case      # same products, another summation order at most


def test_f16_weight_cache_is_per_tensor_object(hip):
    """two different weights that the allocator places at the same address must not share packed planes"""
    H = hip
    g = torch.Generator().manual_seed(2)
    x = _inputs("act", (2, 128, 64, 64), g)
    H.set_f16x2(True)
    ys = []
    for i in range(3):
        w = _cl((torch.randn(128, 128, 3, 3, generator=g) * 0.05).cuda())
        y = H.conv_forward(x, w, None, None, 1, 1)
        ref = F.conv2d(x[:1].double(), w.double(), None, 1, 1)
        assert (y[:1].double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item(), i
        ys.append(y)
        del w
    H.set_f16x2(False)


def _outlier_input(shape, g, kind="act"):
    """ONE element 10^8 x the rest (profiles/r02_precision_f16x2.txt's weak spot): every other value sits below the range of the
    fp16 low term when the tensor is scaled by its maximum"""
    x = _inputs(kind, shape, g)
    x[shape[0] // 2, 17, shape[2] // 2 - 2, shape[3] // 2 - 1] = 1.0e8
    return x


def _rel_err(y, ref, den):
    """error of every output against fp64, relative to ITS OWN sum |a||b| (the scale of its rounding)"""
    return ((y.double() - ref).abs() / den).max().item()


# N, Cin, H, W, Cout, k, kernel family the shape runs on (its F16_STATS counter)
GUARD_CASES = [(8, 256, 64, 64, 256, 3, "conv"),       # tap-strip kernel, TW = 64
               (2, 256, 64, 64, 256, 3, "pg"),         # few-tile 3x3: plane-fed implicit GEMM, 64-row tiles, four K groups per block (round 5)
               (2, 256, 128, 128, 256, 3, "conv"),     # tap-strip kernel, TW = 128, un-split
               (2, 1024, 64, 64, 256, 1, "tiled"),     # tiled kernel 128 x 64
               (8, 1024, 64, 64, 256, 1, "tiled"),     # tiled kernel 128 x 128
               (2, 512, 32, 32, 512, 3, "pg"),         # plane-fed implicit GEMM, K ranges across blocks meeting in one launch
               (2, 256, 64, 64, 1024, 1, "tiled"),     # row-resident 1x1 kernel (K = 256)
               (4, 64, 128, 128, 256, 1, "tiled")]     # row-resident 1x1 kernel (K = 64)


@pytest.mark.parametrize("shape", GUARD_CASES)
def test_range_guard_first_occurrence_on_the_device(hip, shape):
    """the range decision of the fp16 split is taken ON THE DEVICE, per tensor, unlagged.  A tensor with one
    element 10^8 x the rest, seen for the FIRST time by a site that knows nothing (no warm-up call, statistics never flushed to
    the host), comes out at <= 3e-6 of every output's own sum |a||b|: each block reads the producer's statistics slot (max,
    sampled mean) before its first instruction of arithmetic and computes its tile with exact fp32 products
    (csrc/conv_igemm.hip::conv_slow_tile).  The host has not switched anything (fallback counter unchanged, the launch counts
    as an fp16-split launch); an ordinary tensor right afterwards runs the fast path: bit-equal to a library that never saw
    the outlier.  With residual + ReLU epilogue, recorded output statistics included."""
    H = hip
    N, C, Hh, W, Co, k, kind = shape
    g = torch.Generator().manual_seed(31 + sum(shape[:6]))
    x = _outlier_input((N, C, Hh, W), g)
    x2 = _inputs("act", (N, C, Hh, W), g)
    w = _cl((torch.randn(Co, C, k, k, generator=g) * 0.05).cuda())
    sc, sh = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.randn(Co, generator=g) * 0.1).cuda()
    res = _inputs("signed50", (N, Co, Hh, W), g) * 1e-3
    H.set_f16x2(True)          # (clears every site's state)
    c0, f0 = H.F16_STATS[kind], H.F16_STATS["fallback"]
    y = H.conv_forward(x, w, sc, sh, 1, k // 2, relu=True, res=res, res_mode=1)
    assert H.F16_STATS[kind] == c0 + 1 and H.F16_STATS["fallback"] == f0
    xd, wd = x[:1].double(), w.double()
    lin = F.conv2d(xd, wd, None, 1, k // 2) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res[:1].double()
    den = F.conv2d(xd.abs(), wd.abs(), None, 1, k // 2) * sc.double().view(1, -1, 1, 1) + res[:1].double().abs() + 1e-30
    assert _rel_err(y[:1], lin.relu(), den) < 3e-6
    # the statistics the slow path recorded for ITS output are those of the tensor
    slot = y._mmt_amax[0]
    torch.cuda.synchronize()
    assert float(slot.pool.dev[slot.idx, 0]) == float(y.abs().max())
    # an ordinary tensor right behind it: the fast path, bit for bit what a fresh library gives
    y2 = H.conv_forward(x2, w, sc, sh, 1, k // 2, relu=True, res=res, res_mode=1)
    H.set_f16x2(True)
    y2b = H.conv_forward(x2, w, sc, sh, 1, k // 2, relu=True, res=res, res_mode=1)
    assert torch.equal(y2, y2b)
    lin2 = F.conv2d(x2[:1].double(), wd, None, 1, k // 2) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res[:1].double()
    assert (y2[:1].double() - lin2.relu()).abs().max().item() < 1e-5 * lin2.abs().max().item()


@pytest.mark.parametrize("k", [3, 1])
def test_range_guard_data_gradient_and_weight_gradient(hip, k):
    """the same for a backward pass whose incoming gradient carries the outlier (a gradient spike in real training is exactly that
    tensor): the data gradient -- weights exist only as packed, flipped, BN-scaled planes; the slow path reads the forward weight
    through the same flip -- with the ReLU mask in the epilogue, and the weight gradient (dy the bad operand)."""
    H = hip
    from maskrcnn_benchmark.layers import fused
    g = torch.Generator().manual_seed(90 + k)
    N, C, S, Co = 2, 256, 64, 256
    w = _cl((torch.randn(Co, C, k, k, generator=g) * 0.02).cuda())
    bn = (torch.rand(Co, generator=g) + 0.5).cuda()
    dy = _outlier_input((N, Co, S, S), g, "signed50")
    act = _inputs("act", (N, C, S, S), g)
    H.set_f16x2(True)
    f0 = H.F16_STATS["fallback"]
    dx = fused._dgrad(dy, w, (N, C, S, S), 1, k // 2, bn, mask=act)
    assert H.F16_STATS["fallback"] == f0
    wd = (w.double() * bn.double().view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1)
    ref = F.conv2d(dy[:1].double(), wd, None, 1, k // 2) * (act[:1] > 0)
    den = F.conv2d(dy[:1].double().abs(), wd.abs(), None, 1, k // 2) + 1e-30
    assert _rel_err(dx[:1], ref, den) < 3e-6
    # weight gradient: x ordinary, dy with the outlier; both carry statistics slots as producing launches attach them
    x = _inputs("act", (N, C, S, S), g)
    x._mmt_amax = H._amax_of(x)
    dy._mmt_amax = H._amax_of(dy)
    n0 = H.F16_STATS["wgrad"]
    dw = _cl(torch.zeros((Co, C, k, k), device="cuda"))
    db = torch.zeros((Co,), device="cuda")
    H.conv_wgrad(x, dy, (Co, C, k, k), 1, k // 2, dw, bn, db)
    assert H.F16_STATS["wgrad"] == n0 + 1 and H.F16_STATS["fallback"] == f0
    xu = F.unfold(x.double(), k, padding=k // 2)
    refw = torch.einsum("nco,nko->ck", dy.double().flatten(2), xu).view(Co, C, k, k) * bn.double().view(-1, 1, 1, 1)
    denw = torch.einsum("nco,nko->ck", dy.double().abs().flatten(2), xu.abs()).view(Co, C, k, k) * bn.double().view(-1, 1, 1, 1) + 1e-30
    assert _rel_err(dw, refw, denw) < 3e-6
    refb = dy.double().sum((0, 2, 3))
    assert (db.double() - refb).abs().max().item() < 1e-5 * dy.double().abs().sum((0, 2, 3)).max().item()


@pytest.mark.parametrize("bad", ["dy", "x"])
def test_range_guard_of_the_plane_fed_weight_gradient(hip, bad):
    """mmt_conv_wgrad_planes behind the range guard: one operand with an element 10^8 x the rest -- its planes are useless (everything
    else sits below the low term's range) -- and every block computes its part of the tile with exact fp32 products from the fp32
    tensors instead (csrc/conv_wgpl.hip, the `slow` branch): <= 3e-6 of every output's own sum |a||b|, bias gradient included, on the
    first occurrence"""
    H = hip
    g = torch.Generator().manual_seed(17 + len(bad))
    N, C, S, Co, k = 2, 128, 64, 256, 3
    x = _outlier_input((N, C, S, S), g) if bad == "x" else _inputs("act", (N, C, S, S), g)
    dy = _outlier_input((N, Co, S, S), g, "signed50") if bad == "dy" else _inputs("grad", (N, Co, S, S), g)
    bn = (torch.rand(Co, generator=g) + 0.5).cuda()
    H.set_f16x2(True)
    try:
        for t in (x, dy):
            t._mmt_amax = H._amax_of(t)
            H.f16_split_pg(t)
        n0, f0 = H.F16_STATS.get("wgrad_pl", 0), H.F16_STATS["fallback"]
        dw = _cl(torch.zeros((Co, C, k, k), device="cuda"))
        db = torch.zeros((Co,), device="cuda")
        H.conv_wgrad(x, dy, (Co, C, k, k), 1, k // 2, dw, bn, db)
        assert H.F16_STATS.get("wgrad_pl", 0) == n0 + 1 and H.F16_STATS["fallback"] == f0
        xu = F.unfold(x.double(), k, padding=k // 2)
        refw = torch.einsum("nco,nko->ck", dy.double().flatten(2), xu).view(Co, C, k, k) * bn.double().view(-1, 1, 1, 1)
        denw = torch.einsum("nco,nko->ck", dy.double().abs().flatten(2), xu.abs()).view(Co, C, k, k) * bn.double().view(-1, 1, 1, 1) + 1e-30
        assert _rel_err(dw, refw, denw) < 3e-6
        refb = dy.double().sum((0, 2, 3))
        assert (db.double() - refb).abs().max().item() < 1e-5 * dy.double().abs().sum((0, 2, 3)).max().item()
    finally:
        H.set_f16x2(False)


@pytest.mark.parametrize("shape", [(8, 256, 64, 64, 256, 3), (2, 256, 64, 64, 512, 1)])
def test_host_moves_a_persistently_bad_site_to_bf16x3(hip, shape):
    """the slow path is for first occurrences: when the statistics have reached the host the consuming site sees the crest factor
    max / mean > 2^17 itself and runs the 3-term bf16 split (fast, range-free, bit-identical to set_f16x2(False)) until tensors
    with an ordinary range come by again.  Strip kernel (3x3) and tiled kernel (1x1)."""
    H = hip
    N, C, S, _, Co, k = shape
    g = torch.Generator().manual_seed(77 + k)
    x = _outlier_input((N, C, S, S), g)
    x2 = _inputs("act", (N, C, S, S), g)
    w = _cl((torch.randn(Co, C, k, k, generator=g) * 0.05).cuda())
    H.set_f16x2(False)
    y_ref = H.conv_forward(x, w, None, None, 1, k // 2)
    H.set_f16x2(True)
    kind = "conv" if k == 3 else "tiled"
    c0, f0 = H.F16_STATS[kind], H.F16_STATS["fallback"]
    y1 = H.conv_forward(x, w, None, None, 1, k // 2)      # nothing known about this site's inputs yet: device-side slow path
    y1b = H.conv_forward(x, w, None, None, 1, k // 2)     # (the site now waits for the statistics of this tensor)
    assert H.F16_STATS[kind] == c0 + 2 and H.F16_STATS["fallback"] == f0
    assert torch.equal(y1, y1b)
    H.f16_flush_stats()
    y2 = H.conv_forward(x, w, None, None, 1, k // 2)
    assert H.F16_STATS["fallback"] == f0 + 1 and H.F16_STATS[kind] == c0 + 2
    assert torch.equal(y2, y_ref)                          # the host's fall-back IS the 3-term bf16 split
    ref = F.conv2d(x[:1].double(), w.double(), None, 1, k // 2)
    den = F.conv2d(x[:1].double().abs(), w.double().abs(), None, 1, k // 2)
    assert _rel_err(y1[:1], ref, den) < 3e-6 and _rel_err(y2[:1], ref, den) < 3e-6
    # ordinary tensors again: after their statistics arrive the site returns to the fp16 split
    H.conv_forward(x2, w, None, None, 1, k // 2)           # still falling back; the amax pass of the fallback-less path is
    x2._mmt_amax = H._amax_of(x2)                          # not run there, so record the statistics the way a producer would
    H.conv_forward(x2, w, None, None, 1, k // 2)
    H.f16_flush_stats()
    c1 = H.F16_STATS[kind]
    y3 = H.conv_forward(x2, w, None, None, 1, k // 2)
    assert H.F16_STATS[kind] == c1 + 1
    ref3 = F.conv2d(x2[:1].double(), w.double(), None, 1, k // 2)
    assert (y3[:1].double() - ref3).abs().max().item() < 1e-5 * ref3.abs().max().item()


@pytest.mark.parametrize("n", [2, 3, 4])
def test_sum_stats_and_fork(hip, n):
    """mmt_sum_stats: the sum of n gradients in the order given (bit-equal to the chained library additions), its statistics in
    the slot (max exact; mean from the sample within 2 %); ForkFn: the gradient of a tensor with n consumers is that sum, it
    carries the statistics (no reduction pass at its fp16-split consumer), and with one live consumer it is handed through."""
    H = hip
    from maskrcnn_benchmark.layers import fused
    H.set_f16x2(True)
    g = torch.Generator().manual_seed(5 + n)
    ts = [_inputs("grad", (2, 256, 64, 64), g) for _ in range(n)]
    y = H.sum_stats(ts)
    ref = ts[0] + ts[1]
    for t in ts[2:]:
        ref = ref + t
    assert torch.equal(y, ref)
    slot = y._mmt_amax[0]
    H.f16_flush_stats()
    row = slot.pool.host[slot.idx]
    assert float(row[0]) == ref.abs().max().item()
    mean = float(row[1:17].sum()) / float(row[17:33].sum())
    assert abs(mean - ref.abs().mean().item()) < 0.02 * ref.abs().mean().item()
    # the autograd node
    x = _inputs("act", (2, 256, 64, 64), g).requires_grad_(True)
    mid = x * 1.0                                   # a non-leaf: its gradient passes through a hook
    seen = []
    mid.register_hook(lambda gr: seen.append(gr))
    outs = fused.fork(mid, n)
    assert len(outs) == n and all(o.data_ptr() == mid.data_ptr() for o in outs)
    sum(((o * t).sum() for o, t in zip(outs, ts)), torch.zeros((), device="cuda")).backward()
    assert torch.equal(x.grad, ref)
    assert getattr(seen[0], "_mmt_amax", None) is not None and seen[0]._mmt_amax[1] == seen[0]._version
    a0 = H.F16_STATS["amax_pass"]
    w = _cl((torch.randn(256, 256, 3, 3, generator=g) * 0.05).cuda())
    H.conv_forward(seen[0], w, None, None, 1, 1)
    assert H.F16_STATS["amax_pass"] == a0
    # one live consumer: no launch, the gradient itself comes back
    x.grad = None
    mid = x * 1.0
    outs = fused.fork(mid, n)
    (outs[1] * ts[1]).sum().backward()
    assert torch.equal(x.grad, ts[1])


def test_detector_f16x2_vs_oracle(hip):
    """supervised forward + backward of the detector with the strip convolutions on the fp16 split: losses against the fp32
    CPU oracle at 1e-4, parameter gradients against the default arithmetic's run"""
    H = hip
    import synthetic
    from oracle import model as om
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.utils.replay import Replay
    SIZE = 256   # P2 = 64 x 64: few-tile maps -- every convolution on the tiled / row-resident kernels of the fp16 split
    cfg = make_default_cfg()
    torch.manual_seed(0)
    student = build_detection_model(cfg, is_student=True)
    shapes = {k: tuple(v.shape) for k, v in student.state_dict().items()}
    weights = synthetic.make_weights(shapes, seed=0)
    student.load_state_dict(weights, strict=False)
    student.cuda().train()
    imgs, tgs = synthetic.make_labeled(2, SIZE, 4, seed=1234)
    otg = [om.Boxes(t["boxes"], t["size"], {"labels": t["labels"], "masks": t["polys"]}) for t in tgs]
    taps = {}
    torch.manual_seed(99)
    with torch.no_grad():
        ref = om.forward_supervised(weights, om.default_cfg(), imgs, otg, taps)
    ptg = []
    for t in tgs:
        b = BoxList(t["boxes"].cuda(), t["size"], "xyxy")
        b.add_field("labels", t["labels"].cuda())
        b.add_field("masks", SegmentationMask([[p for p in inst] for inst in t["polys"]], t["size"], mode="poly"))
        ptg.append(b)
    grads, losses = {}, {}
    used = []
    orig = H.f16_split
    try:
        for on in (False, True):
            H.set_f16x2(on)
            n0 = H.F16_STATS["tiled"] + H.F16_STATS["conv"] + H.F16_STATS["pg"]
            H.f16_split = lambda x, site=None: (used.append(tuple(x.shape)), orig(x, site))[1]
            for p in student.parameters():
                p.grad = None
            student.set_replay(Replay(taps))
            out = student(to_image_list(list(imgs.cuda()), 32), ptg)
            student.set_replay(None)
            sum(out.values()).backward()
            launched = H.F16_STATS["tiled"] + H.F16_STATS["conv"] + H.F16_STATS["pg"] - n0
            losses[on] = {k: v.item() for k, v in out.items()}
            grads[on] = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
    finally:
        H.f16_split = orig
        H.set_f16x2(False)
    assert launched >= 100, launched   # the forward and data-gradient launches went through the fp16 path
    for k in ref:
        assert losses[True][k] == pytest.approx(ref[k].item(), rel=1e-4, abs=1e-6), (k, losses[True][k], ref[k].item())
    worst = 0.0
    for n, g3 in grads[False].items():
        d = (grads[True][n] - g3).norm().item() / max(g3.norm().item(), 1e-12)
        worst = max(worst, d)
    assert worst < 2e-3, worst   # the tolerance of the gradient comparisons against the oracle (tests/test_model_gpu.py)
