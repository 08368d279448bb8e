"""conv3x3_strip_kernel on the two-term fp16 split, kernel alone (split pass and weight planes prepared once): time vs K at a
fixed tile count -- slope = cost of a super-step (3 taps x 16 channels) of a tile, intercept = everything of a tile outside its
main loop (block dispatch, address setup, first copies, epilogue).  Round 4 (profiles/r04_strip_k_scaling.txt): 1.70 us per
super-step against 1.11 us of MFMA issue time at 2.08 GHz (1.39 us with the copies compiled out), 18-30 us per tile outside
the loop."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as H
L = H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(f, it=10):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def run(x, w, scale, shift, pre):
    N, Cin, Hh, W = x.shape; Cout = w.shape[0]
    (xp, sx), (wp, sw) = pre
    a = H._conv_args(x, w, 1, 1, Hh, W)
    y = H.empty_nhwc(N, Cout, Hh, W, x.device)
    a.y, a.scale, a.shift, a.relu = y.data_ptr(), H._p(scale), H._p(shift), 1
    a.x_planes, a.x_plane_stride, a.w_planes, a.w_plane_stride = xp.data_ptr(), xp.stride(0), wp.data_ptr(), wp.stride(0)
    H._check(L.mmt_conv3x3_strip_f16x2(ctypes.byref(a), sx.data_ptr(), sw.data_ptr(), H._stream()), "strip f16x2")
    return y


g = torch.Generator().manual_seed(0)
for N, Hh in ((8, 256), (2, 256), (8, 128)):
    row = []
    for Cin in (128, 256, 512, 1024):
        x = cl(torch.randn((N, Cin, Hh, Hh), generator=g).relu_().cuda())
        w = cl((torch.randn(256, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda())
        sc = (torch.rand(256, generator=g) + 0.5).cuda(); sh = (torch.randn(256, generator=g) * 0.1).cuda()
        pre = (H.f16_split(x), H.f16_weight_planes(w))
        row.append(timeit(lambda: run(x, w, sc, sh, pre)))
        del x, pre
    tiles = N * Hh * Hh // 256 * 2
    per_cu = tiles / 256.0
    slope = (row[3] - row[1]) / (3 * (1024 - 256) / 16) / per_cu * 1e3     # us per super-step of a tile
    icpt = (row[1] - slope * (3 * 256 / 16) * per_cu / 1e3) / per_cu * 1e3   # us per tile outside the loop
    print("N=%d %d^2  ms at Cin 128/256/512/1024: %s | %.3f us per super-step, %.1f us per tile outside the loop (%d tiles, %.0f per CU)" % (
        N, Hh, " ".join("%.3f" % t for t in row), slope, icpt, tiles, per_cu), flush=True)
