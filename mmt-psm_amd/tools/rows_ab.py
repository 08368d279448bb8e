"""The row-resident 1x1 kernel against the tiled kernel (MMT_ROWS=0) on the step's shapes, in isolation: bit equality of
the outputs, us per launch and the minimum-traffic rate of the row-resident one (profiles/r04_rows_epilogue.txt).

    python mmt-psm_amd/tools/rows_ab.py
"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maskrcnn_benchmark import _hip as H  # noqa: E402

# (N, H, W, Cin, Cout, residual mode, mask)
SHAPES = [(8, 64, 64, 256, 1024, 0, 0), (8, 64, 64, 256, 1024, 1, 0), (4, 64, 64, 256, 1024, 1, 0), (2, 64, 64, 256, 1024, 1, 0),
          (8, 128, 128, 128, 512, 1, 0), (4, 128, 128, 128, 512, 1, 0), (8, 256, 256, 64, 256, 1, 0),
          (4, 256, 256, 64, 256, 1, 0), (4, 64, 64, 256, 1024, 1, 1), (4, 256, 256, 256, 256, 2, 0),
          (4, 256, 256, 64, 256, 0, 1), (4, 256, 256, 256, 64, 1, 1), (8, 256, 256, 256, 64, 0, 0)]


def cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":
        H.LIB_PATH = os.path.abspath(sys.argv[2])
    H.set_f16x2(True)
    print("%-34s %9s %9s %7s  equal" % ("shape", "tiled us", "rows us", "GB/s"))
    for (N, Hh, W, Cin, Cout, rm, mk) in SHAPES:
        g = torch.Generator().manual_seed(N + Cin + Cout)
        x, w = cl(torch.randn(N, Cin, Hh, W, generator=g)), cl(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05)
        sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
        res = None
        if rm == 1:
            res = cl(torch.randn(N, Cout, Hh, W, generator=g))
        elif rm == 2:
            res = cl(torch.randn(N, Cout, Hh // 2, W // 2, generator=g))
        mask = cl(torch.randn(N, Cout, Hh, W, generator=g)) if mk else None
        out, t = {}, {}
        for d in ("0", "1"):
            os.environ["MMT_ROWS"] = d
            kw = dict(relu=not mk, res=res, res_mode=rm)
            if mk:
                kw.update(mask=mask)
            f = lambda: H.conv_forward(x, w, sc, sh, 1, 0, **kw)
            out[d] = f()
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[d] = e0.elapsed_time(e1) / 20 * 1e3
        M = N * Hh * W
        byt = M * Cin * 4 + M * Cout * 4 + (res.numel() * 4 if res is not None else 0) + (M * Cout * 4 if mk else 0)
        print("%-34s %9.1f %9.1f %7.0f  %s" % (str((N, Hh, W, Cin, Cout, rm, mk)), t["0"], t["1"], byt / t["1"] / 1e3,
                                               torch.equal(out["0"], out["1"])))


if __name__ == "__main__":
    main()
