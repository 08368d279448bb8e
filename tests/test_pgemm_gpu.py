"""The plane-fed implicit GEMM of round 5 (include/mmtpsm.h: mmt_conv_forward_pg; csrc/conv_pgemm.hip): the default arithmetic
(two-term fp16 split, 3 products per multiply) for any (KH, KW, stride, pad) with both operands as pre-split fp16 planes.

  * same products in the same order as the tiled kernel behind mmt_conv_forward_f16x2: BIT-IDENTICAL outputs for an equal number of
    K ranges -- every epilogue operand (FrozenBN scale / shift, bias, residual, ReLU, the ReLU mask of a data gradient), both tile
    heights, ragged M and Cout, strided and 1x1 forms, the fc shape;
  * against an fp64 convolution: within the default mode's tolerance (tests/test_hip_kernels.py MODE_TOL[3] = 1e-5 of max |ref|);
  * K ranges side by side INSIDE a block (128-row tiles: 2, 64-row tiles: 4 groups of waves whose accumulators meet in LDS) and
    across blocks in ONE launch (partial tiles meet in the per-stream workspace, the last arriver of a tile adds them in fixed
    order): the same ranges summed in the same order whichever way they are distributed -- bit-identical among themselves;
    deterministic (two launches agree bit for bit, counters left at zero), equal to the un-split sum to rounding, the library's own
    plan included; several launches back to back on one stream and two streams at once;
  * the data-gradient form (weights that exist only as flipped / transposed / BN-scaled planes) against layers.fused._dgrad;
  * the fp16 split's range guard: a tensor whose crest factor defeats fp16 takes exact fp32 products (<= 3e-6 of sum |a||b|);
  * the recorded statistics of the output (max |y|, sampled mean) are those of the tensor."""
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
    H.set_f16x2(None)
    H.set_conv_precision(prev)
    os.environ.pop("MMT_SPLITK", None)
    os.environ.pop("MMT_STRIP", None)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _make(case, seed=0):
    N, C, Hh, W, Co, k, stride, pad, opts = case
    g = torch.Generator().manual_seed(seed + N + C + Hh + Co + k)
    x = _cl(torch.randn(N, C, Hh, W, generator=g).relu().cuda())
    w = _cl((torch.randn(Co, C, k, k, generator=g) * (2.0 / (k * k * C)) ** 0.5).cuda())
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    sc = (torch.rand(Co, generator=g) + 0.5).cuda() if "bn" in opts else None
    sh = (torch.randn(Co, generator=g) * 0.1).cuda() if ("bn" in opts or "bias" in opts) else None
    res = _cl(torch.randn(N, Co, Ho, Wo, generator=g).cuda()) if "res" in opts else None
    mask = _cl(torch.randn(N, Co, Ho, Wo, generator=g).cuda()) if "mask" in opts else None
    kw = dict(relu="relu" in opts, res=res, mask=mask, mask_scale=2.0)
    return x, w, sc, sh, kw


def _ref64(x, w, sc, sh, kw, stride, pad, rows=1):
    ref = F.conv2d(x[:rows].double(), w.double(), None, stride, pad)
    if sc is not None:
        ref = ref * sc.double().view(1, -1, 1, 1)
    if sh is not None:
        ref = ref + sh.double().view(1, -1, 1, 1)
    if kw["res"] is not None:
        ref = ref + kw["res"][:rows].double()
    if kw["relu"]:
        ref = F.relu(ref)
    if kw["mask"] is not None:
        ref = torch.where(kw["mask"][:rows] > 0, ref * kw["mask_scale"], torch.zeros_like(ref))
    return ref


CASES = [  # N, Cin, H, W, Cout, k, stride, pad, epilogue
    (2, 256, 64, 64, 256, 3, 1, 1, "bn relu"),          # layer3 conv2 of a student pass
    (2, 512, 32, 32, 512, 3, 1, 1, "bn relu"),          # layer4 conv2
    (2, 256, 32, 32, 256, 3, 1, 1, "bias"),             # FPN output conv on P5
    (3, 256, 16, 16, 256, 3, 1, 1, "bias relu"),        # RPN head on P6: M = 768
    (25, 256, 14, 14, 256, 3, 1, 1, "bias relu"),       # mask head, ragged M (4900)
    (2, 128, 64, 64, 128, 3, 1, 1, "mask"),             # a data gradient's ReLU mask
    (2, 1024, 32, 32, 256, 1, 1, 0, "bn relu"),         # 1x1 with long K
    (2, 512, 32, 32, 2048, 1, 1, 0, "bn res relu"),     # layer4 conv3 + residual
    (2, 256, 64, 64, 192, 3, 1, 1, "bn"),               # Cout not a multiple of 128
    (2, 64, 40, 24, 96, 3, 2, 1, "bias relu"),          # stride 2, odd sizes, Cout < 128
    (2, 512, 64, 64, 256, 1, 2, 0, "bn"),               # strided 1x1 (first block of a stage)
    (300, 1024, 1, 1, 1024, 1, 1, 0, "bias relu"),      # fc7
    (2, 32, 24, 24, 64, 7, 1, 3, "res"),                # 7x7, residual without affine
]


def _fits(case, rows, ks):
    """does (tile rows, K ranges across blocks) fit the library's limits?  (64 MiB of partial tiles per stream, >= 1 step per range)"""
    N, C, Hh, W, Co, k, stride, pad, _ = case
    M = N * ((Hh + 2 * pad - k) // stride + 1) * ((W + 2 * pad - k) // stride + 1)
    tiles = ((M + rows - 1) // rows) * ((Co + 127) // 128)
    return ks * (256 // rows) <= (C * k * k) // 16 and (ks == 1 or tiles * ks * rows * 128 * 4 <= (64 << 20))


@pytest.mark.parametrize("case", CASES)
def test_bit_identical_to_the_tiled_kernel_and_close_to_fp64(hip, case):
    H = hip
    N, C, Hh, W, Co, k, stride, pad, opts = case
    x, w, sc, sh, kw = _make(case)
    os.environ["MMT_SPLITK"] = "0"      # one K range in the tiled kernel
    os.environ["MMT_STRIP"] = "0"       # ... and never the tap-strip kernel (same products, another summation order)
    os.environ["MMT_ROWS"] = "0"
    os.environ["MMT_PG"] = "0"          # ... nor this kernel: the reference is the tiled kernel
    try:
        y_old = H.conv_forward(x, w, sc, sh, stride, pad, relu=kw["relu"], res=kw["res"], res_mode=1 if kw["res"] is not None else 0,
                               mask=kw["mask"], mask_scale=kw["mask_scale"])
    finally:
        os.environ.pop("MMT_ROWS", None)
        os.environ.pop("MMT_PG", None)
    xp = H.f16_split(x)
    y = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=256, ksplit=1, xp=xp, **kw)
    assert y.shape == y_old.shape
    assert torch.equal(y, y_old), (case, (y - y_old).abs().max().item())
    ref = _ref64(x, w, sc, sh, kw, stride, pad)
    scale = ref.abs().max().item()
    assert (y[:1].double() - ref).abs().max().item() / scale < 1e-5, case
    # the statistics slot: max |y| exactly, the sampled mean within a factor of two of the tensor's
    torch.cuda.synchronize()
    slot = y._mmt_amax[0]
    st = slot.pool.dev[slot.idx].cpu()
    assert st[0].item() == y.abs().max().item()
    if st[17:33].sum() > 0:
        mean = (st[1:17].sum() / st[17:33].sum()).item()
        assert 0.5 * y.abs().mean().item() <= mean <= 2.0 * y.abs().mean().item() + 1e-12
    # the shorter tiles run 2 / 4 K ranges side by side inside the block: the sums of that many ranges across blocks, bit for bit
    for rows in (128, 64):
        if not _fits(case, 256, 256 // rows):
            continue
        yk = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=rows, ksplit=1, xp=xp, **kw)
        yr = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=256, ksplit=256 // rows, xp=xp, **kw)
        assert torch.equal(yk, yr), (case, rows, (yk - yr).abs().max().item())
        assert (yk[:1].double() - ref).abs().max().item() / scale < 1e-5, (case, rows)
        assert yk._mmt_amax[0].pool.dev[yk._mmt_amax[0].idx][0].item() == yk.abs().max().item()


SPLIT_CASES = [
    ((2, 256, 64, 64, 256, 3, 1, 1, "bn relu"), (2, 4, 9)),
    ((2, 512, 32, 32, 512, 3, 1, 1, "mask"), (3, 8, 16)),
    ((2, 2048, 32, 32, 512, 1, 1, 0, "bn res relu"), (2, 5)),
    ((64, 12544, 1, 1, 1024, 1, 1, 0, "bias relu"), (7, 16)),     # fc6 on few rows: one row tile
    ((25, 256, 14, 14, 256, 3, 1, 1, "bias relu"), (2, 3)),
]


@pytest.mark.parametrize("case,splits", SPLIT_CASES)
def test_split_k_in_one_launch(hip, case, splits):
    H = hip
    N, C, Hh, W, Co, k, stride, pad, opts = case
    x, w, sc, sh, kw = _make(case, seed=5)
    xp = H.f16_split(x)
    y1 = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=256, ksplit=1, xp=xp, **kw)
    ref = _ref64(x, w, sc, sh, kw, stride, pad, rows=min(N, 2))
    scale = ref.abs().max().item()
    ran = 0
    for ks in splits:
        for rows in (64, 128, 256):
            if not _fits(case, rows, ks):
                continue
            ran += 1
            ya = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=rows, ksplit=ks, xp=xp, **kw)
            yb = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=rows, ksplit=ks, xp=xp, **kw)
            assert torch.equal(ya, yb), (case, ks, rows)                      # fixed summation order, counters back at zero
            err = (ya[:min(N, 2)].double() - ref).abs().max().item() / scale
            assert err < 1e-5, (case, ks, rows, err)
            assert (ya - y1).abs().max().item() <= 4e-6 * scale + 1e-30, (case, ks, rows)
    assert ran >= 3
    y0 = H.conv_forward_pg(x, w, sc, sh, stride, pad, xp=xp, **kw)       # the library's own plan
    assert (y0[:min(N, 2)].double() - ref).abs().max().item() / scale < 1e-5


def test_split_k_equals_the_tiled_kernels_split_k(hip):
    """equal K ranges, equal order of the partial sums: the launch pair of the tiled kernel and the one-launch form agree bit for bit"""
    H = hip
    import ctypes
    case = (2, 256, 64, 64, 256, 3, 1, 1, "bn relu")
    N, C, Hh, W, Co, k, stride, pad, opts = case
    x, w, sc, sh, kw = _make(case, seed=11)
    os.environ["MMT_STRIP"] = "0"
    os.environ["MMT_PG"] = "0"
    try:
        y_old = H.conv_forward(x, w, sc, sh, stride, pad, relu=True)
    finally:
        os.environ.pop("MMT_PG", None)
    a = H.ConvArgs()
    a.x = a.w_planes = 16
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW = N, Hh, W, C, Co, k, k
    a.stride, a.pad, a.Ho, a.Wo, a.out_stride = stride, pad, Hh, W, 1
    ks = H.lib().mmt_conv_ksplit(ctypes.byref(a))
    assert ks > 1
    y = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=256, ksplit=ks, **kw)       # every range in a block of its own
    assert torch.equal(y, y_old)
    if ks == 4:                                                                            # ... or all four inside one block
        assert torch.equal(H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=64, ksplit=1, **kw), y_old)


def test_back_to_back_and_two_streams(hip):
    """the workspace and its counters belong to a stream: launches of one stream follow each other, two streams run side by side"""
    H = hip
    case = (2, 256, 64, 64, 256, 3, 1, 1, "bn relu")
    N, C, Hh, W, Co, k, stride, pad, opts = case
    xs = [_make(case, seed=s) for s in range(4)]
    want = [H.conv_forward_pg(x, w, sc, sh, stride, pad, ksplit=1, **kw) for x, w, sc, sh, kw in xs]
    planes = [H.f16_split(t[0]) for t in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None] * 4
    for rep in range(3):
        for i, (x, w, sc, sh, kw) in enumerate(xs):
            with torch.cuda.stream(streams[i & 1]):
                got[i] = H.conv_forward_pg(x, w, sc, sh, stride, pad, ksplit=4, xp=planes[i], **kw)
        torch.cuda.synchronize()
        for i in range(4):
            scale = want[i].abs().max().item()
            assert (got[i] - want[i]).abs().max().item() <= 4e-6 * scale, (rep, i)


@pytest.mark.parametrize("shape", [(2, 256, 64, 64, 256, 3, 1, 1), (2, 512, 32, 32, 2048, 1, 1, 0), (2, 256, 64, 64, 512, 1, 2, 0)])
def test_data_gradient_form(hip, shape):
    """dx = conv(dy, flip / transpose / BN-scale of w) * (act > 0): the weights exist only as packed planes"""
    H = hip
    from maskrcnn_benchmark.layers import fused
    N, Cin, Hh, W, Cout, k, stride, pad = shape
    if stride != 1:
        pytest.skip("strided data gradients scatter (out_stride): not a shape of this kernel")
    g = torch.Generator().manual_seed(sum(shape))
    w = _cl((torch.randn(Cout, Cin, k, k, generator=g) * 0.02).cuda())
    bn = (torch.rand(Cout, generator=g) + 0.5).cuda()
    dy = _cl((torch.randn(N, Cout, Hh, W, generator=g) * torch.exp(torch.randn(N, Cout, Hh, W, generator=g) * 2.0) * 1e-5).cuda())
    act = _cl(torch.randn(N, Cin, Hh, W, generator=g).relu().cuda())
    os.environ["MMT_SPLITK"] = "0"
    os.environ["MMT_STRIP"] = "0"
    os.environ["MMT_ROWS"] = "0"
    os.environ["MMT_PG"] = "0"
    try:
        want = fused._dgrad(dy, w, (N, Cin, Hh, W), stride, pad, bn, mask=act)
    finally:
        os.environ.pop("MMT_ROWS", None)
        os.environ.pop("MMT_PG", None)
    got = H.conv_forward_pg(dy, None, None, None, 1, k - 1 - pad, mask=act, mask_scale=1.0, w_shape=(Cin, Cout, k, k),
                            f16_src=(w, bn), tile_rows=256, ksplit=1)
    assert torch.equal(got, want), (got - want).abs().max().item()
    wd = (w.double() * bn.double().view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1)
    ref = F.conv2d(dy[:1].double(), wd, None, 1, k - 1 - pad) * (act[:1] > 0)
    assert (got[:1].double() - ref).abs().max().item() / ref.abs().max().item() < 1e-5


@pytest.mark.parametrize("rows,ksplit", [(256, 1), (64, 1), (128, 4)])
def test_range_guard_takes_exact_products(hip, rows, ksplit):
    """one element 1e8 x the rest: everything else falls below the low fp16 term; the block sees it in the statistics slot of ITS
    operand and computes the tile with fp32 products (first occurrence, no warm-up call)"""
    H = hip
    g = torch.Generator().manual_seed(3)
    N, C, S, Co = 2, 256, 32, 256
    x = torch.randn(N, C, S, S, generator=g).relu()
    x[0, 3, 5, 7] = 1e8
    x = _cl(x.cuda())
    w = _cl((torch.randn(Co, C, 3, 3, generator=g) * 0.02).cuda())
    y = H.conv_forward_pg(x, w, None, None, 1, 1, tile_rows=rows, ksplit=ksplit)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    bound = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    assert ((y.double() - ref).abs() / bound.clamp_min(1e-30)).max().item() < 6e-6   # (a chain of 2304 fp32 FMAs)


def test_the_dispatcher_takes_it_for_3x3_on_small_maps(hip):
    """conv_forward / the data gradient of layers.fused run the library's choice: 3x3 convolutions that are not tap-strip shapes"""
    H = hip
    from maskrcnn_benchmark.layers import fused
    case = (2, 256, 32, 32, 256, 3, 1, 1, "bn relu")
    x, w, sc, sh, kw = _make(case, seed=2)
    n0 = H.F16_STATS["pg"]
    y = H.conv_forward(x, w, sc, sh, 1, 1, relu=True)
    y2 = H.conv_forward(x, w, sc, sh, 1, 1, relu=True)            # the recorded launch plan
    assert H.F16_STATS["pg"] == n0 + 2
    assert torch.equal(y, y2) and torch.equal(y, H.conv_forward_pg(x, w, sc, sh, 1, 1, relu=True))
    dx = fused._dgrad(y, w, x.shape, 1, 1, sc, mask=x)
    assert H.F16_STATS["pg"] == n0 + 3
    wd = (w.double() * sc.double().view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1)
    ref = F.conv2d(y[:1].double(), wd, None, 1, 1) * (x[:1] > 0)
    assert (dx[:1].double() - ref).abs().max().item() / ref.abs().max().item() < 1e-5
    os.environ["MMT_PG"] = "0"
    try:
        y3 = H.conv_forward(x, w, sc, sh, 1, 1, relu=True)
    finally:
        os.environ.pop("MMT_PG", None)
    assert H.F16_STATS["pg"] == n0 + 3 and (y3 - y).abs().max().item() <= 4e-6 * y.abs().max().item()
    # a 1x1 layer and a tap-strip shape stay where they were
    H.conv_forward(_cl(torch.randn(2, 1024, 32, 32).cuda()), _cl(torch.randn(256, 1024, 1, 1).cuda() * 0.03))
    H.conv_forward(_cl(torch.randn(8, 128, 128, 128).cuda()), _cl(torch.randn(128, 128, 3, 3).cuda() * 0.03), None, None, 1, 1)
    assert H.F16_STATS["pg"] == n0 + 3


def test_library_plan(hip):
    H = hip
    assert H.conv_pg_plan(2, 256, 64, 64, 256, 3, 3, 1, 1) == (64, 1)
    assert H.conv_pg_plan(400, 256, 14, 14, 256, 3, 3, 1, 1) == (256, 1)
    assert H.conv_pg_plan(2, 512, 32, 32, 512, 3, 3, 1, 1) == (64, 2)
    assert H.conv_pg_plan(2, 3, 64, 64, 64, 7, 7, 2, 3) == (0, 0)          # Cin % 16 != 0
    assert H.conv_pg_plan(2, 256, 64, 64, 15, 1, 1, 1, 0) == (0, 0)        # Cout <= 32


@pytest.mark.parametrize("case", CASES)
def test_row_blocked_planes_equal_planes_indexed_like_x(hip, case):
    """the input planes in the row-blocked order [N H][C / 16][W][16] (mmt_split_planes_f16_rb; what the step uses: runs of up to
    1 KiB per copy instruction) are a re-ordering of the planes indexed like x, and the kernel returns the same bits from either"""
    H = hip
    N, C, Hh, W, Co, k, stride, pad, opts = case
    x, w, sc, sh, kw = _make(case)
    xp0 = H.f16_split(x)
    assert H.PG_RB
    xp1 = H.f16_split_pg(x)
    assert xp1[2] == 1
    a = xp0[0].view(2, N, Hh, W, C // 16, 16).permute(0, 1, 2, 4, 3, 5).contiguous().view(2, -1)
    assert torch.equal(a, xp1[0])
    assert torch.equal(xp0[1][:1], xp1[1][:1])
    for rows in (256, 128, 64):
        if not _fits(case, rows, 1):
            continue
        y0 = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=rows, ksplit=1, xp=xp0, **kw)
        y1 = H.conv_forward_pg(x, w, sc, sh, stride, pad, tile_rows=rows, ksplit=1, xp=xp1, **kw)
        assert torch.equal(y0, y1), (case, rows)
    # the library's own choice of tile and K ranges, the split done inside the call
    y2 = H.conv_forward_pg(x, w, sc, sh, stride, pad, **kw)
    y3 = H.conv_forward_pg(x, w, sc, sh, stride, pad, xp=xp0, **kw)
    assert torch.equal(y2, y3), case


def test_product_library_has_no_copy_wave_split_form(hip):
    """round 6: the x_planes = NULL form (copy waves split fp32 rows themselves; round-5 experiment, slower, 144 registers of copy
    ring = spills) is compiled into the tools build only (`make ablate`): the product library refuses it loudly"""
    H = hip
    x, w, sc, sh, kw = _make(CASES[0])
    N, C, Hh, W, Co, k, stride, pad, opts = CASES[0]
    with pytest.raises(RuntimeError):
        H.conv_forward_pg(x, w, sc, sh, stride, pad, xp="fp32", **kw)
