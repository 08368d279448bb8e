"""EXPERIMENT: the tap-strip 3x3 kernel on a two-term fp16 split (3 matrix products per multiply, include/mmtpsm.h:
mmt_conv3x3_strip_f16x2) against the shipped three-term bf16 split (6 products) and the fp32-input MFMA kernel:
time, and error against an fp64 convolution, for activation-like and gradient-like operands."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as H
L = H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def f16x2(x, w, scale, shift, relu, pre=None):
    """the kernel alone on prepared planes (pre) or with its preparation passes (amax + split of x; the weight planes are
    cached per weight version as in the model)"""
    N, Cin, Hh, W = x.shape; Cout = w.shape[0]
    xp, sx = pre[0] if pre is not None else H.f16_split(x)
    wp, sw = pre[1] if pre is not None else H.f16_weight_planes(w)
    a = H._conv_args(x, w, 1, 1, Hh, W)
    y = H.empty_nhwc(N, Cout, Hh, W, x.device)
    a.y, a.scale, a.shift, a.relu = y.data_ptr(), H._p(scale), H._p(shift), 1 if relu else 0
    a.x_planes, a.x_plane_stride, a.w_planes, a.w_plane_stride = xp.data_ptr(), xp.stride(0), wp.data_ptr(), wp.stride(0)
    H._check(L.mmt_conv3x3_strip_f16x2(ctypes.byref(a), sx.data_ptr(), sw.data_ptr(), H._stream()), "strip f16x2")
    return y, ((xp, sx), (wp, sw))
H.lib()
H.set_f16x2(False)   # tool: the arms below switch the arithmetic explicitly
g = torch.Generator().manual_seed(0)
def act(shape): return cl(torch.randn(shape, generator=g).relu_().cuda())                       # post-ReLU activations
def grad(shape): return cl((torch.randn(shape, generator=g) * torch.exp(torch.randn(shape, generator=g) * 2.0) * 1e-5).cuda())  # 6 decades
def signed50(shape): return cl((torch.randn(shape, generator=g) * 50.0).cuda())
def tiny(shape): return cl((torch.randn(shape, generator=g) * 1e-20).cuda())
def huge(shape): return cl((torch.randn(shape, generator=g) * 1e15).cuda())
def outl(shape):   # 1 value in 10^4 is 10^5 x the rest
    t = torch.randn(shape, generator=g) * 1e-3
    m = torch.rand(shape, generator=g) < 1e-4
    return cl(torch.where(m, t * 1e5, t).cuda())
def outl8(shape):   # ONE value 10^8 x the rest
    t = torch.randn(shape, generator=g) * 1e-3
    t.view(-1)[12345] = 1e5
    return cl(t.cuda())
cases = [("fpn 3x3 256@256^2 N8, activations", 8, 256, 256, 256, 256, act), ("fpn 3x3 256@256^2 N2, activations", 2, 256, 256, 256, 256, act),
         ("rpn 3x3 256@128^2 N8, activations", 8, 256, 128, 128, 256, act), ("l2 3x3 128@128^2 N8, activations", 8, 128, 128, 128, 128, act),
         ("fpn 3x3 256@256^2 N2, GRADIENT-like input (1e-5 x lognormal)", 2, 256, 256, 256, 256, grad),
         ("l3 3x3 256@64^2 N8, signed x 50", 8, 256, 64, 64, 256, signed50), ("l3 3x3 256@64^2 N2 (split-K), gradient-like", 2, 256, 64, 64, 256, grad),
         ("rpn 3x3 256@128^2 N2, 1e-20 scale", 2, 256, 128, 128, 256, tiny), ("rpn 3x3 256@128^2 N2, 1e15 scale", 2, 256, 128, 128, 256, huge),
         ("rpn 3x3 256@128^2 N2, outliers 1e5 x (1 in 1e4)", 2, 256, 128, 128, 256, outl),
         ("rpn 3x3 256@128^2 N2, ONE outlier 1e8 x", 2, 256, 128, 128, 256, outl8)]
for name, N, Cin, Hh, W, Cout, mk in cases:
    x = mk((N, Cin, Hh, W)); w = cl((torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda())
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda()
    sh = (torch.randn(Cout, generator=g) * 0.1).cuda() * {grad: 1e-5, tiny: 1e-20, huge: 1e15, outl: 1e-3, outl8: 1e-3}.get(mk, 1.0)
    fl = 2.0 * N * Hh * W * Cout * Cin * 9
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    refabs = torch.nn.functional.conv2d(x[:1].double().abs(), w.double().abs(), None, 1, 1) * sc.double().view(1, -1, 1, 1)   # sum |a||b|: the scale of each output's rounding
    H.set_conv_precision(3)
    xp3 = H.split_planes(x)
    y3 = H.conv_forward(x, w, sc, sh, 1, 1, x_planes=xp3)
    t3 = timeit(lambda: H.conv_forward(x, w, sc, sh, 1, 1, x_planes=xp3))
    H.set_conv_precision(0)
    y0 = H.conv_forward(x, w, sc, sh, 1, 1)
    t0 = timeit(lambda: H.conv_forward(x, w, sc, sh, 1, 1), 5)
    H.set_conv_precision(3)
    yh, pre = f16x2(x, w, sc, sh, False)
    th = timeit(lambda: f16x2(x, w, sc, sh, False, pre))
    tp = timeit(lambda: H.f16_split(x))
    H.set_f16x2(True)
    ym = H.conv_forward(x, w, sc, sh, 1, 1)          # the product's own call in this mode
    H.set_f16x2(False)
    assert torch.equal(ym, yh)
    def err(y):
        d = (y[:1].double() - ref).abs()
        return (d.max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                (d / refabs.clamp_min(1e-300)).max().item())
    e0, e3, eh = err(y0), err(y3), err(yh)
    print("%s\n   fp32-input MFMA %7.3f ms %6.1f TF  err max/max %.2e rms %.2e per-output max|d|/sum|a||b| %.2e\n   bf16 x3 (6 prod) %7.3f ms %6.1f TF  err max/max %.2e rms %.2e per-output %.2e\n"
          "   fp16 x2 (3 prod) %7.3f ms %6.1f TF  err max/max %.2e rms %.2e per-output %.2e   [+ amax / split passes of the input %.3f ms; scales 2^%d, 2^%d]" % (
              name, t0, fl / t0 / 1e9, e0[0], e0[1], e0[2], t3, fl / t3 / 1e9, e3[0], e3[1], e3[2], th, fl / th / 1e9, eh[0], eh[1], eh[2], tp,
              round(math.log2(pre[0][1][0].item())), round(math.log2(pre[1][1][0].item()))))
