"""the five legacy names of `maskrcnn_benchmark._C` (csrc/vision.cpp:7-13) and the operator API built on
them (layers/nms.py:5, layers/roi_align.py:11-67, structures/boxlist_ops.py:9-35) "must remain and remain correct on their
own" -- exercised here on the GPU through exactly those names against the reference's fixtures (tests/golden/nms.npz,
roi_align.npz: outputs of the reference's own `_C` CPU build) and the oracle's ROIAlign backward.  The model itself uses
the batched / fused entry points; nothing below goes through them.

Also: the route-B stub of INTEGRATION.md (the file a maintainer of the reference adds as `maskrcnn_benchmark/_C.py`) is
extracted from the document and executed verbatim."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import gold, T, ROOT, PKG

pytestmark = pytest.mark.gpu


def _nms_cases():
    g = gold("nms")
    return g, [i for i in range(6)]


def test_layers_nms_golden():
    from maskrcnn_benchmark import layers, _C
    assert layers.nms is _C.nms
    g, cases = _nms_cases()
    for i in cases:
        keep = layers.nms(T(g["b%d" % i]).cuda(), T(g["s%d" % i]).cuda(), float(g["t%d" % i]))
        assert keep.dtype == torch.int64 and keep.is_cuda           # csrc/cuda/nms.cu:127-130: device-resident indices
        np.testing.assert_array_equal(keep.cpu().numpy(), g["k%d" % i])
    e = layers.nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.5)
    assert e.numel() == 0 and e.dtype == torch.int64              # csrc/nms.h:17-18


def test_boxlist_nms_golden():
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.boxlist_ops import boxlist_nms
    g, _ = _nms_cases()
    b, s = T(g["b3"]).cuda(), T(g["s3"]).cuda()     # 2000 score-sorted boxes, the RPN's call (rpn/inference.py:130-135)
    bl = BoxList(b, (600, 600), "xyxy")
    bl.add_field("objectness", s)
    out = boxlist_nms(bl, float(g["t3"]), max_proposals=300, score_field="objectness")
    k = g["k3"][:300]
    np.testing.assert_array_equal(out.bbox.cpu().numpy(), g["b3"][k])
    np.testing.assert_array_equal(out.get_field("objectness").cpu().numpy(), g["s3"][k])
    # xywh list comes back in its own mode (boxlist_ops.py:24-35); threshold <= 0 returns the list itself (:22-23)
    xywh = bl.convert("xywh")
    out2 = boxlist_nms(xywh, float(g["t3"]), score_field="objectness")
    assert out2.mode == "xywh" and len(out2) == len(g["k3"])
    assert boxlist_nms(bl, 0, score_field="objectness") is bl


def _roi_cases():
    g = gold("roi_align")
    return g, int(g["n"])


def test_C_roi_align_forward_golden():
    from maskrcnn_benchmark import _C
    g, n = _roi_cases()
    for i in range(n):
        sc, ph, pw, sr = g["p%d" % i]
        y = _C.roi_align_forward(T(g["x%d" % i]).cuda(), T(g["r%d" % i]).cuda(), float(sc), int(ph), int(pw), int(sr))
        assert tuple(y.shape) == tuple(g["y%d" % i].shape)
        np.testing.assert_array_equal(y.cpu().numpy(), g["y%d" % i])   # bit-exact vs the reference's CPU kernel


def test_layers_roi_align_autograd_and_module():
    """layers.roi_align / layers.ROIAlign: the reference's autograd contract (forward + once_differentiable backward,
    gradient for the input only) with NCHW tensors in and out, as poolers.py:66,119 calls them"""
    from maskrcnn_benchmark import layers
    from oracle import native
    g, n = _roi_cases()
    for i in (1, 2, 3, 5):
        sc, ph, pw, sr = g["p%d" % i]
        x = T(g["x%d" % i]).cuda().requires_grad_()
        rois = T(g["r%d" % i]).cuda()
        y = layers.roi_align(x, rois, (int(ph), int(pw)), float(sc), int(sr))
        np.testing.assert_array_equal(y.detach().cpu().numpy(), g["y%d" % i])
        gen = torch.Generator().manual_seed(40 + i)
        gy = torch.randn(y.shape, generator=gen)
        y.backward(gy.cuda())
        ref = native.roi_align_backward(gy, T(g["r%d" % i]), float(sc), int(ph), int(pw), *g["x%d" % i].shape, int(sr))
        assert x.grad.shape == x.shape
        assert (x.grad.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        assert rois.grad is None
    m = layers.ROIAlign((7, 7), 0.25, 2)
    assert repr(m) == "ROIAlign(output_size=(7, 7), spatial_scale=0.25, sampling_ratio=2)"   # layers/roi_align.py:60-67
    y = m(T(g["x1"]).cuda(), T(g["r1"]).cuda())
    np.testing.assert_array_equal(y.cpu().numpy(), g["y1"])


def test_C_errors_and_roi_pool():
    from maskrcnn_benchmark import _C, layers
    with pytest.raises(RuntimeError):
        _C.roi_pool_forward(torch.zeros(1, 1, 4, 4).cuda(), torch.zeros(1, 5).cuda(), 1.0, 2, 2)
    with pytest.raises(RuntimeError):
        layers.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)     # CPU tensors are refused: no CPU path in the product


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Keep the reference package"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "route-B stub not found in INTEGRATION.md"
    return m.group(1)


def test_route_b_stub_verbatim():
    """INTEGRATION.md route B: the ctypes `_C` a maintainer drops into the REFERENCE tree, executed as printed (only the
    library path is filled in), then driven the way the reference's layers/nms.py and layers/roi_align.py drive `_C`"""
    src = _stub_source()
    assert "/path/to/libmmtpsm.so" in src
    ns = {}
    exec(compile(src.replace("/path/to/libmmtpsm.so", os.path.join(PKG, "libmmtpsm.so")), "INTEGRATION.md:route-B", "exec"), ns)
    for name in ("nms", "roi_align_forward", "roi_align_backward", "roi_pool_forward", "roi_pool_backward"):
        assert callable(ns[name]), name                              # csrc/vision.cpp:7-13
    g, _ = _nms_cases()
    for i in range(6):
        keep = ns["nms"](T(g["b%d" % i]).cuda(), T(g["s%d" % i]).cuda(), float(g["t%d" % i]))
        np.testing.assert_array_equal(keep.cpu().numpy(), g["k%d" % i])
    assert ns["nms"](torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.5).numel() == 0

    class _ROIAlign(torch.autograd.Function):      # the reference's wrapper, layers/roi_align.py:11-44, over the stub
        @staticmethod
        def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
            ctx.save_for_backward(roi)
            ctx.cfg = (output_size, spatial_scale, sampling_ratio, input.size())
            return ns["roi_align_forward"](input, roi, spatial_scale, output_size[0], output_size[1], sampling_ratio)

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, grad_output):
            rois, = ctx.saved_tensors
            (ph, pw), sc, sr, (bs, ch, h, w) = ctx.cfg
            return ns["roi_align_backward"](grad_output, rois, sc, ph, pw, bs, ch, h, w, sr), None, None, None, None

    from oracle import native
    g, n = _roi_cases()
    for i in (0, 1, 2, 5):
        sc, ph, pw, sr = g["p%d" % i]
        x = T(g["x%d" % i]).cuda().requires_grad_()     # plain NCHW-contiguous input, as the reference hands it over
        y = _ROIAlign.apply(x, T(g["r%d" % i]).cuda(), (int(ph), int(pw)), float(sc), int(sr))
        np.testing.assert_array_equal(y.detach().cpu().numpy(), g["y%d" % i])
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(70 + i))
        y.backward(gy.cuda())
        ref = native.roi_align_backward(gy, T(g["r%d" % i]), float(sc), int(ph), int(pw), *g["x%d" % i].shape, int(sr))
        assert (x.grad.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(RuntimeError):
        ns["roi_pool_forward"]()
