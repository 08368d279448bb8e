"""EXPERIMENT: the DMA-fed tiled kernel (1x1 layers, 3x3 on small maps) on the two-term fp16 split, activations split in
registers after the copy (include/mmtpsm.h: mmt_conv_forward_f16x2), against the default arithmetic: time and error vs fp64."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as H
L = H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
def timeit(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def f16(x, w, sc, sh, stride, pad, relu, res, amax, wp, sw):
    N, Cin, Hh, W = x.shape; Cout, _, k, _ = w.shape
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    a = H._conv_args(x, w, stride, pad, Ho, Wo)
    y = H.empty_nhwc(N, Cout, Ho, Wo, x.device)
    a.y, a.scale, a.shift, a.relu = y.data_ptr(), H._p(sc), H._p(sh), 1 if relu else 0
    if res is not None:
        a.res, a.res_mode = res.data_ptr(), 1
    a.w_planes, a.w_plane_stride = wp.data_ptr(), wp.stride(0)
    H._check(L.mmt_conv_forward_f16x2(ctypes.byref(a), amax.data_ptr(), sw.data_ptr(), H._stream()), "conv f16x2")
    return y
H.lib()
H.set_f16x2(False)   # tool: the arms below switch the arithmetic explicitly
g = torch.Generator().manual_seed(0)
cases = [("1x1 256->1024 @64^2 N8 +res+relu", 8, 256, 64, 1024, 1, 1, True), ("1x1 1024->256 @64^2 N8 relu", 8, 1024, 64, 256, 1, 1, False),
         ("1x1 256->1024 @64^2 N2 +res+relu", 2, 256, 64, 1024, 1, 1, True), ("1x1 1024->256 @64^2 N2 relu", 2, 1024, 64, 256, 1, 1, False),
         ("1x1 512->128 @128^2 N8 relu", 8, 512, 128, 128, 1, 1, False), ("1x1 256->256 @256^2 N8 (FPN lateral)", 8, 256, 256, 256, 1, 1, False),
         ("1x1 2048->512 @32^2 N8 relu", 8, 2048, 32, 512, 1, 1, False), ("3x3 512->512 @32^2 N8 relu", 8, 512, 32, 512, 3, 1, False),
         ("1x1 s2 512->1024 @128^2 N8", 8, 512, 128, 1024, 1, 2, False), ("fc 12544->1024, 1024 rows", 1024, 12544, 1, 1024, 1, 1, False)]
for name, N, Cin, S, Cout, k, stride, has_res in cases:
    x = cl(torch.randn(N, Cin, S, S, generator=g).relu_().cuda())
    w = cl((torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).cuda())
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda(); sh = (torch.randn(Cout, generator=g) * 0.1).cuda()
    pad = k // 2
    So = (S + 2 * pad - k) // stride + 1
    res = cl(torch.randn(N, Cout, So, So, generator=g).cuda()) if has_res else None
    kw = dict(relu=True, res=res, res_mode=1 if has_res else 0)
    fl = 2.0 * N * So * So * Cout * Cin * k * k
    y3 = H.conv_forward(x, w, sc, sh, stride, pad, **kw)
    t3 = timeit(lambda: H.conv_forward(x, w, sc, sh, stride, pad, **kw))
    amax = x.abs().max().reshape(1)
    wp, sw = H.f16_weight_planes(w)
    yh = f16(x, w, sc, sh, stride, pad, True, res, amax, wp, sw)
    th = timeit(lambda: f16(x, w, sc, sh, stride, pad, True, res, amax, wp, sw))
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), None, stride, pad) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if has_res: ref = ref + res[:1].double()
    ref = ref.relu()
    scale = ref.abs().max().item()
    e3 = (y3[:1].double() - ref).abs().max().item() / scale; eh = (yh[:1].double() - ref).abs().max().item() / scale
    r3 = ((y3[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(); rh = ((yh[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print("%-38s default %7.3f ms %6.1f TF err %.2e rms %.2e | fp16 x2 %7.3f ms %6.1f TF err %.2e rms %.2e | x%.2f" % (
        name, t3, fl / t3 / 1e9, e3, r3, th, fl / th / 1e9, eh, rh, t3 / th))
