"""conv3x3_strip_kernel (fp16 split): time vs K at a fixed tile count -- slope = cost of a super-step, intercept = prologue +
epilogue + launch of a tile; variants of tools/bench_strip_stages.py (MMT_STRIP_VARIANT)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as H
from bench_strip_stages import run, timeit, cl   # noqa (runs that tool's table first when imported as a script is avoided below)
g = torch.Generator().manual_seed(0)
for N, Hh in ((8, 256), (2, 256), (8, 128)):
    for var in ("0", "1", "11"):
        os.environ["MMT_STRIP_VARIANT"] = var
        row = []
        for Cin in (128, 256, 512, 1024):
            x = cl(torch.randn((N, Cin, Hh, Hh), generator=g).relu_().cuda())
            w = cl((torch.randn(256, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda())
            sc = (torch.rand(256, generator=g) + 0.5).cuda(); sh = (torch.randn(256, generator=g) * 0.1).cuda()
            pre = (H.f16_split(x), H.f16_weight_planes(w))
            row.append(timeit(lambda: run(x, w, sc, sh, pre), it=10))
            del x, pre
        tiles = N * Hh * Hh // 256 * 2
        per_cu = tiles / 256.0
        slope = (row[3] - row[1]) / (3 * (1024 - 256) / 16) / per_cu * 1e3     # us per super-step of a tile
        icpt = (row[1] - slope * (3 * 256 / 16) * per_cu / 1e3) / per_cu * 1e3   # us per tile outside the loop
        print("N=%d %d^2 variant %-2s  ms at Cin 128/256/512/1024: %s | %.3f us per super-step, %.1f us per tile outside the loop (%d tiles, %.0f per CU)" % (
            N, Hh, var, " ".join("%.3f" % t for t in row), slope, icpt, tiles, per_cu), flush=True)
