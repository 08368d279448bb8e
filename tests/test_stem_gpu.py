"""The ResNet stem as ONE launch (include/mmtpsm.h: mmt_stem_fused; csrc/conv_stem.hip): conv 7x7 / stride 2 / pad 3 + FrozenBatchNorm
+ ReLU + max pool 3x3 / stride 2 / pad 1 (reference modeling/backbone/resnet.py:288-293, layers/batch_norm.py:19-24).

  * bit-identical to the three-launch stem it replaces (space-to-depth copy, tiled convolution, maxpool_kernel): same products in
    the same order, the same maxima -- full tiles, ragged right / bottom edges, maps smaller than one tile;
  * against an fp64 formulation of the reference's stem: the default mode's tolerance;
  * the recorded statistics of the output are the tensor's;
  * an image whose dynamic range defeats fp16 (one pixel 1e8 x the rest) takes exact fp32 products on the device."""
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
def stem():
    from maskrcnn_benchmark import _hip as H
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.backbone import backbone as B
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(3)
    H.set_f16x2(True)
    g = torch.Generator().manual_seed(3)
    m = B.StemWithFixedBatchNorm(make_default_cfg())
    with torch.no_grad():
        m.conv1.weight.copy_(torch.randn(m.conv1.weight.shape, generator=g) * 0.05)
        m.bn1.weight.copy_(torch.rand(64, generator=g) + 0.5)
        m.bn1.bias.copy_(torch.randn(64, generator=g) * 0.2)
        m.bn1.running_mean.copy_(torch.randn(64, generator=g) * 0.1)
        m.bn1.running_var.copy_(torch.rand(64, generator=g) + 0.5)
    m.cuda()
    yield H, B, m
    B._STEM_FUSED[0] = True
    H.set_f16x2(None)
    H.set_conv_precision(prev)


def _ref64(m, x):
    s, b = m.bn1.folded()
    y = F.conv2d(x.double(), m.conv1.weight.double(), None, 2, 3) * s.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    return F.max_pool2d(F.relu(y), 3, 2, 1)


@pytest.mark.parametrize("shape", [(2, 256, 256), (1, 1024, 1024), (2, 160, 200), (3, 32, 36), (1, 8, 8), (2, 100, 60)])
def test_fused_stem_equals_the_three_launch_stem(stem, shape):
    H, B, m = stem
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(n, 3, h, w, generator=g) * 60.0).cuda()        # mean-subtracted BGR-255 pixels
    n0 = H.C_CALLS[0]
    B._STEM_FUSED[0] = True
    with torch.no_grad():
        y = m(x)
    calls = H.C_CALLS[0] - n0
    B._STEM_FUSED[0] = False
    with torch.no_grad():
        y3 = m(x)
    assert y.shape == y3.shape == (n, 64, h // 4, w // 4)
    assert torch.equal(y, y3), (y - y3).abs().max().item()
    assert calls <= 4                                               # statistics pass of x, (first call: weight packing,) the launch
    ref = _ref64(m, x)
    assert (y.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    torch.cuda.synchronize()
    slot = y._mmt_amax[0]
    st = slot.pool.dev[slot.idx].cpu()
    assert st[0].item() == y.abs().max().item()
    mean = (st[1:17].sum() / st[17:33].sum()).item()
    assert 0.5 * y.mean().item() <= mean <= 2.0 * y.mean().item()


def test_fused_stem_range_guard(stem):
    H, B, m = stem
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 64, 96, generator=g) * 60.0
    x[1, 2, 17, 40] = 6e9
    x = x.cuda()
    with torch.no_grad():
        y = m(x)
    s, b = m.bn1.folded()
    lin = F.conv2d(x.double(), m.conv1.weight.double(), None, 2, 3)
    den = F.conv2d(x.double().abs(), m.conv1.weight.double().abs(), None, 2, 3) * s.double().abs().view(1, -1, 1, 1) + b.double().abs().view(1, -1, 1, 1)
    ref = F.max_pool2d(F.relu(lin * s.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)), 3, 2, 1)
    bound = F.max_pool2d(den, 3, 2, 1)
    assert ((y.double() - ref).abs() / bound).max().item() < 3e-6


@pytest.mark.parametrize("shape,opts", [((2, 256, 256), "bn relu"), ((8, 64, 64), "bn relu"), ((2, 40, 40), "bias"), ((3, 24, 56), "relu"),
                                        ((1, 8, 16), "bn"), ((2, 13, 21), "bn relu")])
def test_layer1_3x3_patch_kernel_equals_the_tiled_kernel(stem, shape, opts):
    """conv 3x3, 64 -> 64 channels (+ FrozenBN + ReLU): layer1's conv2 on conv3x3_c64_kernel (csrc/conv_stem.hip) -- one gather of the
    input patch per 8 x 16 output tile instead of nine trips through the L2 -> LDS path; same products in the same order as the tiled
    kernel it replaces: bit-identical, ragged maps included"""
    H, B, m = stem
    n, h, w_ = shape
    g = torch.Generator().manual_seed(sum(shape) + len(opts))
    x = torch.randn(n, 64, h, w_, generator=g).relu().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).cuda().contiguous(memory_format=torch.channels_last)
    sc = (torch.rand(64, generator=g) + 0.5).cuda() if "bn" in opts else None
    sh = (torch.randn(64, generator=g) * 0.1).cuda() if ("bn" in opts or "bias" in opts) else None
    y = H.conv_forward(x, w, sc, sh, 1, 1, relu="relu" in opts)
    os.environ["MMT_C64"] = "0"
    try:
        y_old = H.conv_forward(x, w, sc, sh, 1, 1, relu="relu" in opts)
    finally:
        os.environ.pop("MMT_C64", None)
    assert torch.equal(y, y_old), (y - y_old).abs().max().item()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    if sc is not None:
        ref = ref * sc.double().view(1, -1, 1, 1)
    if sh is not None:
        ref = ref + sh.double().view(1, -1, 1, 1)
    if "relu" in opts:
        ref = ref.relu()
    assert (y.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    torch.cuda.synchronize()
    slot = y._mmt_amax[0]
    assert slot.pool.dev[slot.idx, 0].item() == y.abs().max().item()


def test_layer1_3x3_patch_kernel_range_guard(stem):
    H, B, m = stem
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 24, 40, generator=g).relu()
    x[1, 7, 3, 30] = 4e8
    x = x.cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.06).cuda().contiguous(memory_format=torch.channels_last)
    H.set_f16x2(True)
    y = H.conv_forward(x, w, None, None, 1, 1)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    bound = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    assert ((y.double() - ref).abs() / bound.clamp_min(1e-30)).max().item() < 3e-6
