"""bf16 STORAGE of activations in the bf16 arithmetic mode (BASELINE configs[4] "bf16 MFMA path"; include/mmtpsm.h:
mmt_conv_args.io_bf16).  The statement tested: a convolution whose operands / residual / mask / result are bf16 TENSORS
computes what the fp32-storage call computes on the same (bf16-representable) values, rounded to nearest even at the end --
mode 1 rounds its fp32 operands to bf16 anyway, so storing them rounded changes nothing but the bytes moved.

  * forward / data-gradient kernels (all-planes kernels with one term, the tap-strip kernel, split-K + finish launch):
    against the fp32-storage call on x.float(), every epilogue form of the ResNet body (BN, ReLU, residual, ReLU mask,
    strided scatter);
  * weight gradient with bf16 x and / or bf16 dy: the same launch on the widened tensors (same kernel, the load widens
    exactly; compared up to the order of the bias atomics);
  * max-pool: bit-equal;
  * one bottleneck, forward + backward, bf16 storage vs fp32 storage;
  * the detector: supervised losses against the fp32 CPU oracle at the tolerances of the fp32-storage bf16 test
    (tests/test_irnet_gpu.py::test_irnet_bf16_products_mode), and the body really holds bf16 tensors."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
sys.path.insert(0, ROOT)

BF = torch.bfloat16


@pytest.fixture()
def mode1():
    from maskrcnn_benchmark import _hip as H
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(1)
    yield H
    H.set_bf16_storage(False)
    H.set_conv_precision(prev)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rand(shape, g, scale=1.0):
    return _cl((torch.randn(shape, generator=g) * scale).cuda())


CASES = [
    # N, Cin, H, W, Cout, k, stride, pad                 what runs
    (2, 256, 32, 32, 64, 1, 1, 0),      # 1x1 reduce: 128 x 64 tiles, K = 256
    (2, 64, 32, 32, 256, 1, 1, 0),      # 1x1 expand, K = 64
    (2, 256, 32, 32, 128, 1, 2, 0),     # strided 1x1 (STRIDE_IN_1X1)
    (2, 64, 40, 40, 64, 3, 1, 1),       # 3x3 with 64 channels: all-planes kernel (the strip kernel wants Cin >= 128)
    (2, 128, 64, 64, 128, 3, 1, 1),     # 3x3 on the strip kernel, 64-pixel strips (split-K: few tiles)
    (8, 128, 128, 128, 128, 3, 1, 1),   # strip kernel, 128-pixel strips, un-split
    (2, 2048, 16, 16, 512, 1, 1, 0),    # long K, few tiles: split-K + finish launch on the all-planes kernel
    (3, 96, 17, 23, 80, 3, 1, 1),       # ragged: M, Cout not multiples of the tile
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_storage_matches_fp32_storage(mode1, case):
    H = mode1
    N, Cin, Hh, W, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = _rand((N, Cin, Hh, W), g).to(BF)
    w = _rand((Cout, Cin, k, k), g, (2.0 / (Cin * k * k)) ** 0.5)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda() * 0.1
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _rand((N, Cout, Ho, Wo), g).to(BF)
    mask = _rand((N, Cout, Ho, Wo), g).to(BF)
    for kw in (dict(), dict(relu=True, res=res, res_mode=1), dict(mask=mask, mask_scale=1.25, res=res, res_mode=1)):
        wide = {kk: (v.float() if isinstance(v, torch.Tensor) else v) for kk, v in kw.items()}
        ref = H.conv_forward(x.float(), w, scale, shift, stride, pad, **wide)            # fp32 storage, same values
        got32 = H.conv_forward(x, w, scale, shift, stride, pad, **kw)                     # bf16 in, fp32 out
        got16 = H.conv_forward(x, w, scale, shift, stride, pad, out_dtype=BF, **kw)      # bf16 in, bf16 out
        assert got32.dtype == torch.float32 and got16.dtype == BF
        tol = 2e-5 * (Cin * k * k) ** 0.5 + 1e-5   # summation order of the strip / split-K forms
        assert (got32 - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), case
        # the bf16 result is the fp32 result rounded to nearest even: equal to rounding `got32`
        assert torch.equal(got16, got32.to(BF)), case


def test_conv_fp32_input_bf16_output_and_scatter(mode1):
    """the stem (fp32 image -> bf16) and the strided data gradient (bf16 dy scattered into a zeroed bf16 tensor)"""
    H = mode1
    g = torch.Generator().manual_seed(5)
    x = _rand((2, 16, 48, 48), g)
    w = _rand((64, 16, 4, 4), g, 0.1)
    ref = H.conv_forward(x, w, None, None, 1, 2, relu=True, out_size=(48, 48))
    got = H.conv_forward(x, w, None, None, 1, 2, relu=True, out_size=(48, 48), out_dtype=BF)
    assert torch.equal(got, ref.to(BF))
    dy = _rand((2, 128, 16, 16), g).to(BF)
    w1 = _rand((256, 128, 1, 1), g, 0.1)   # data gradient of a stride-2 1x1: 128 -> 256 channels, scattered to 32 x 32
    full = _rand((2, 256, 32, 32), g).to(BF)
    ref = H.conv_forward(dy.float(), w1, out_stride=2, out_hw=(32, 32), mask=full.float())
    got = H.conv_forward(dy, w1, out_stride=2, out_hw=(32, 32), mask=full, out_dtype=BF)
    assert got.dtype == BF and torch.equal(got, ref.to(BF))
    assert got[:, :, 1::2, :].abs().max().item() == 0 and got[:, :, :, 1::2].abs().max().item() == 0


@pytest.mark.parametrize("shape", [(2, 256, 32, 32, 64, 1, 1, 0), (2, 64, 32, 32, 64, 3, 1, 1), (2, 256, 32, 32, 128, 1, 2, 0),
                                   (4, 128, 9, 9, 128, 3, 1, 1)])
def test_wgrad_bf16_storage(mode1, shape):
    H = mode1
    N, Cin, Hh, W, Cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = _rand((N, Cin, Hh, W), g).to(BF)
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = _rand((N, Cout, Ho, Wo), g).to(BF)
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda()

    def run(xx, dd):
        dw = _cl(torch.zeros((Cout, Cin, k, k), device="cuda"))
        db = torch.zeros((Cout,), device="cuda")
        H.conv_wgrad(xx, dd, (Cout, Cin, k, k), stride, pad, dw, rs, db)
        return dw, db
    ref_w, ref_b = run(x.float(), dy.float())
    assert ref_w.abs().max().item() > 0
    for xx, dd in ((x, dy.float()), (x.float(), dy), (x, dy)):
        dw, db = run(xx, dd)
        # split accumulation by atomics: the order of the partial sums differs from launch to launch
        assert (dw - ref_w).abs().max().item() <= 2e-5 * max(1.0, ref_w.abs().max().item())
        assert (db - ref_b).abs().max().item() <= 2e-5 * max(1.0, ref_b.abs().max().item())


def test_maxpool_bf16(mode1):
    H = mode1
    g = torch.Generator().manual_seed(3)
    x = _rand((2, 64, 37, 41), g).to(BF)
    assert torch.equal(H.maxpool3x3s2(x), H.maxpool3x3s2(x.float()).to(BF))


def test_bottleneck_bf16_storage_vs_fp32_storage(mode1):
    """one strided bottleneck with a downsample branch, forward + backward: the bf16-storage run against the fp32-storage
    run of the same (bf16) arithmetic.  Differences: the block's intermediate tensors and gradients are rounded to bf16."""
    H = mode1
    from maskrcnn_benchmark.layers import fused
    g = torch.Generator().manual_seed(11)
    N, Cin, mid, Cout, S = 2, 256, 128, 512, 32
    x32 = _rand((N, Cin, S, S), g).relu_().to(BF).float()
    ws = [_rand((mid, Cin, 1, 1), g, (2.0 / Cin) ** 0.5), _rand((mid, mid, 3, 3), g, (2.0 / (9 * mid)) ** 0.5),
          _rand((Cout, mid, 1, 1), g, (2.0 / mid) ** 0.5), _rand((Cout, Cin, 1, 1), g, (2.0 / Cin) ** 0.5)]
    bn = []
    for c in (mid, mid, Cout, Cout):
        bn += [(torch.rand(c, generator=g) * 0.5 + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()]
    gout = _rand((N, Cout, S // 2, S // 2), g).to(BF).float()
    res = {}
    for storage in (False, True):
        H.set_bf16_storage(storage)
        x = (x32.to(BF) if storage else x32.clone()).requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        out = fused.BottleneckFn.apply(x, wl[0], wl[1], wl[2], wl[3], tuple(bn), 2)
        assert out.dtype == (BF if storage else torch.float32)
        go = gout * (out.detach().float() > 0)   # the convention of layers/fused.py: gradient arrives masked by (out > 0)
        out.backward(go.to(out.dtype))
        assert x.grad.dtype == x.dtype
        res[storage] = (out.detach().float(), x.grad.float(), [w.grad.clone() for w in wl])
    H.set_bf16_storage(False)
    o0, dx0, dw0 = res[False]
    o1, dx1, dw1 = res[True]

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert rel(o1, o0) < 1e-2, rel(o1, o0)      # three roundings to bf16 (2^-9 each) on the way
    assert rel(dx1, dx0) < 2e-2, rel(dx1, dx0)
    for a, b in zip(dw1, dw0):
        assert rel(a, b) < 2e-2, rel(a, b)
    assert rel(o1, o0) > 0   # it is a different run


def test_detector_bf16_storage_vs_oracle(mode1):
    """supervised forward + backward of the detector with bf16 activation storage in the ResNet body: losses against the
    fp32 CPU oracle at the tolerance of the fp32-storage bf16 test, finite gradients in the flat buffer, bf16 body"""
    H = mode1
    import synthetic
    from oracle import model as om
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.utils.replay import Replay
    SIZE = 160
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
    seen = []
    hook = student.backbone.body.layer3.register_forward_hook(lambda m, i, o: seen.append(o.dtype))
    hook2 = student.backbone.register_forward_hook(lambda m, i, o: seen.extend(t.dtype for t in o))   # P2..P6
    H.set_bf16_storage(True)
    try:
        student.set_replay(Replay(taps, substitute_lists=True))
        out = student(to_image_list(list(imgs.cuda()), 32), ptg)
        student.set_replay(None)
        sum(out.values()).backward()
    finally:
        H.set_bf16_storage(False)
        hook.remove()
        hook2.remove()
    assert len(seen) >= 6 and all(d == BF for d in seen), seen
    dev = {k: abs(out[k].item() - ref[k].item()) / max(abs(ref[k].item()), 1e-6) for k in ref}
    assert all(v == v for v in dev.values())
    for k, v in dev.items():
        assert v < 5e-2, dev
    gr = [p.grad for n, p in student.named_parameters() if p.grad is not None and "layer3" in n]
    assert gr and all(torch.isfinite(t).all().item() for t in gr) and any(t.abs().max().item() > 0 for t in gr)


def test_pair_forward_with_bf16_storage(mode1):
    """the N = 4 forward of the two student passes (backbone.py::forward_pair) hands out dense bf16 slices: pyramids equal to
    the batched pass, and a backward through one half reaches the weights"""
    H = mode1
    import synthetic
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.modeling.backbone.backbone import forward_pair
    cfg = make_default_cfg()
    torch.manual_seed(0)
    model = build_detection_model(cfg, is_student=True)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synthetic.make_weights(shapes, seed=0), strict=False)
    model.cuda().train()
    g = torch.Generator().manual_seed(4)
    xa, xb = (torch.randn(2, 3, 192, 192, generator=g) * 50).cuda(), (torch.randn(2, 3, 192, 192, generator=g) * 50).cuda()
    H.set_bf16_storage(True)
    try:
        with torch.no_grad():
            cat = model.backbone(torch.cat([xa, xb], 0))
        pa, pb = forward_pair(model.backbone, xa, xb)
        for c, a, b in zip(cat, pa, pb):
            assert a.dtype == BF and torch.equal(c[:2], a.detach()) and torch.equal(c[2:], b.detach())
        sum(t.float().mean() for t in pa).backward()
    finally:
        H.set_bf16_storage(False)
    w = model.backbone.body.layer3[0].conv2.weight
    assert w.grad is not None and torch.isfinite(w.grad).all().item() and w.grad.abs().max().item() > 0


def test_heads_and_weight_gradients_multiply_in_bf16_in_mode_1(mode1):
    """BASELINE configs[4] "bf16 MFMA path" (VERDICT r5 N1): mode 1 is GLOBAL -- the heads' layers (fc6, the mask head's 3x3 on pooled
    features, the RPN head's 3x3) and every weight gradient multiply bf16-rounded operands with fp32 accumulation like the backbone does;
    only the STORAGE of the heads' activations stays fp32 (they are produced by fp32 ROIAlign / selection kernels).  Shown the way an
    arithmetic can be shown from outside: the result equals the fp64 product of the bf16-ROUNDED operands to fp32-accumulation accuracy
    (<= 2e-5 of sum |a||b|) and lies several times farther from the product of the un-rounded operands -- what bf16 rounding must cost."""
    import torch.nn.functional as F
    H = mode1
    from maskrcnn_benchmark.layers import fused
    g = torch.Generator().manual_seed(5)

    def bf(t):
        return t.to(BF).to(torch.float64)

    def check(got, rounded, exact, bound, what):
        e_r = ((got.double().cpu() - rounded).abs() / bound).max().item()
        e_x = ((got.double().cpu() - exact).abs() / bound).max().item()
        assert e_r <= 2e-5, (what, e_r)
        assert e_x >= 3.0 * e_r and e_x >= 2e-5, (what, e_x, e_r)   # (the rounding's share shrinks like 1 / sqrt(K): 12 544 terms for fc6)

    # fc6-like Linear (R x 12544 -> 1024) through the model's own node
    x = (torch.randn(64, 12544, generator=g)).relu().cuda()
    w = (torch.randn(1024, 12544, generator=g) * 0.01).cuda()
    y = fused.linear(x, w, None, relu=False)
    xc, wc = x.cpu(), w.cpu()
    check(y, bf(xc) @ bf(wc).t(), xc.double() @ wc.double().t(), (xc.double().abs() @ wc.double().abs().t()).clamp_min(1e-300), "fc6")
    # the mask head's 3x3 on pooled features, the RPN head's 3x3 on a small level
    for shape, co in (((25, 256, 14, 14), 256), ((2, 256, 32, 32), 256)):
        x = _cl(torch.randn(*shape, generator=g).relu().cuda())
        w = _cl((torch.randn(co, shape[1], 3, 3, generator=g) * 0.02).cuda())
        y = H.conv_forward(x, w, None, None, 1, 1)
        xc, wc = x.cpu(), w.cpu()
        check(y, F.conv2d(bf(xc), bf(wc), None, 1, 1), F.conv2d(xc.double(), wc.double(), None, 1, 1),
              F.conv2d(xc.double().abs(), wc.double().abs(), None, 1, 1).clamp_min(1e-300), "conv %s" % (shape,))
    # a weight gradient (the mask head's layer; fp32 tensors in, bf16 products)
    x = _cl(torch.randn(25, 256, 14, 14, generator=g).relu().cuda())
    dy = _cl((torch.randn(25, 256, 14, 14, generator=g) * 1e-2).cuda())
    dw = _cl(torch.zeros(256, 256, 3, 3, device="cuda"))
    H.conv_wgrad(x, dy, (256, 256, 3, 3), 1, 1, dw)
    torch.cuda.synchronize()
    xc, dc = x.cpu(), dy.cpu()
    rw = torch.nn.grad.conv2d_weight(bf(xc), (256, 256, 3, 3), bf(dc), stride=1, padding=1)
    ex = torch.nn.grad.conv2d_weight(xc.double(), (256, 256, 3, 3), dc.double(), stride=1, padding=1)
    bd = torch.nn.grad.conv2d_weight(xc.double().abs(), (256, 256, 3, 3), dc.double().abs(), stride=1, padding=1).clamp_min(1e-300)
    check(dw, rw, ex, bd, "wgrad")
