"""Device time per launch of the plane-fed implicit GEMM (csrc/conv_pgemm.hip) against the kernels `conv_forward` dispatches to today,
on the convolution shapes of a step that the tap-strip kernel does not take.  A shape's calls are captured into a hipGraph (REPS calls,
replayed) so that the host's issue time per call -- 25-35 us through ctypes, the floor of every event-bracketed Python loop
(profiles/r04_history.md) -- is not in the number.  Columns: old = what conv_forward runs (tiled kernel, with its split-K launch pair),
pg = the new kernel alone (planes given), pg+split = with the plane-split pass of x in front, per (tile rows, K ranges) variant.

    python mmt-psm_amd/tools/bench_pg.py [--reps 20] [--quick] > gpurun_out/bench_pg.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskrcnn_benchmark import _hip as H  # noqa: E402

SHAPES = [  # N, Cin, H, W, Cout, k, stride, pad, calls per step (forward + data gradient, r04_conv_table.txt)
    (2, 256, 64, 64, 256, 3, 1, 1, 17), (4, 256, 64, 64, 256, 3, 1, 1, 7), (8, 256, 64, 64, 256, 3, 1, 1, 7),
    (2, 512, 32, 32, 512, 3, 1, 1, 6), (4, 512, 32, 32, 512, 3, 1, 1, 3), (8, 512, 32, 32, 512, 3, 1, 1, 3),
    (2, 128, 128, 128, 128, 3, 1, 1, 8), (4, 128, 128, 128, 128, 3, 1, 1, 4),
    (2, 256, 32, 32, 256, 3, 1, 1, 5), (2, 256, 16, 16, 256, 3, 1, 1, 3), (8, 256, 32, 32, 256, 3, 1, 1, 2),
    (400, 256, 14, 14, 256, 3, 1, 1, 4), (25, 256, 14, 14, 256, 3, 1, 1, 8), (100, 256, 14, 14, 256, 3, 1, 1, 4),
    (4096, 12544, 1, 1, 1024, 1, 1, 0, 1), (2000, 12544, 1, 1, 1024, 1, 1, 0, 1), (1024, 12544, 1, 1, 1024, 1, 1, 0, 2),
    (1024, 1024, 1, 1, 12544, 1, 1, 0, 2), (1024, 1024, 1, 1, 1024, 1, 1, 0, 4),
    (2, 1024, 64, 64, 256, 1, 1, 0, 12), (4, 1024, 64, 64, 256, 1, 1, 0, 6), (8, 1024, 64, 64, 256, 1, 1, 0, 6),
    (2, 2048, 32, 32, 512, 1, 1, 0, 6), (8, 2048, 32, 32, 512, 1, 1, 0, 2), (2, 512, 128, 128, 128, 1, 1, 0, 8),
    (2, 512, 32, 32, 2048, 1, 1, 0, 4), (8, 512, 32, 32, 2048, 1, 1, 0, 3), (8, 512, 128, 128, 128, 1, 1, 0, 3),
]
QUICK = [0, 1, 2, 3, 6, 11, 12, 14, 19, 20, 21, 22, 24, 25, 27]


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def timed(fn, reps, stream):
    """ms per call of fn() over `reps` calls replayed from a graph (falls back to a plain loop when capture fails)"""
    with torch.cuda.stream(stream):
        fn()
        fn()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                fn()
        run = g.replay
        mode = "graph"
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()

        def run():
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    fn()
        mode = "loop(%s)" % type(e).__name__
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        best = t if best is None else min(best, t)
    return best, mode


def ablate(args, stream):
    """what the main loop waits for: the same launch with parts of it compiled out (results are wrong by construction)"""
    names = {0: "as shipped", 1: "A copies re-read one step", 2: "B copies re-read one step", 3: "A and B re-read", 4: "no copies (zeros)",
             8: "no fragment reads", 12: "neither", 16: "register-staged copies (correct results)", 32: "A pieces of 64 B", 64: "A pieces of 128 B"}
    for sh, rows in [((2, 256, 64, 64, 256, 3, 1, 1), 64), ((2, 1024, 64, 64, 256, 1, 1, 0), 64), ((400, 256, 14, 14, 256, 3, 1, 1), 256),
                     ((8, 256, 64, 64, 256, 3, 1, 1), 256), ((4, 256, 64, 64, 256, 3, 1, 1), 128), ((2, 512, 32, 32, 512, 3, 1, 1), 128)]:
        N, C, Hh, W, Co, k, s, p = sh
        g = torch.Generator().manual_seed(1)
        x = cl(torch.randn(N, C, Hh, W, generator=g).relu().cuda())
        w = cl((torch.randn(Co, C, k, k, generator=g) * (2.0 / (k * k * C)) ** 0.5).cuda())
        b = torch.randn(Co, generator=g).cuda()
        with torch.cuda.stream(stream):
            H._amax_of(x)
            xp = H.f16_split_pg(x)
        out = []
        y0 = H.conv_forward_pg(x, w, None, b, s, p, relu=True, tile_rows=rows, ksplit=1, xp=xp)
        os.environ["MMT_PG_DBG"] = "16"
        y16 = H.conv_forward_pg(x, w, None, b, s, p, relu=True, tile_rows=rows, ksplit=1, xp=xp)
        os.environ.pop("MMT_PG_DBG")
        torch.cuda.synchronize()
        out.append("register-staged == shipped: %s" % bool(torch.equal(y0, y16)))
        for dbg in ((0, 16, 4) if rows == 128 else (0, 16, 32, 64, 1, 2, 3, 4, 8, 12)):
            if rows == 256 and dbg == 3:
                continue
            os.environ["MMT_PG_DBG"] = str(dbg)
            t, _ = timed(lambda: H.conv_forward_pg(x, w, None, b, s, p, relu=True, tile_rows=rows, ksplit=1, xp=xp), args.reps, stream)
            out.append("%s %.1f" % (names[dbg], t * 1e3))
        os.environ.pop("MMT_PG_DBG", None)
        print("%s rows %d: %s" % (sh, rows, " | ".join(out)), flush=True)


def split_passes(args, stream):
    """the plane-split pass in the two orders (indexed like x / row-blocked): us per launch, GB/s of 8 bytes per element"""
    for sh in [(8, 256, 256, 256), (4, 256, 256, 256), (2, 256, 256, 256), (8, 128, 128, 128), (2, 256, 64, 64), (2, 512, 32, 32),
               (400, 256, 14, 14), (25, 256, 14, 14), (2, 2048, 32, 32)]:
        x = cl(torch.randn(*sh).relu().cuda())
        with torch.cuda.stream(stream):
            H._amax_of(x)
        t0, _ = timed(lambda: H.f16_split(x), args.reps, stream)
        t1, _ = timed(lambda: H.f16_split_pg(x), args.reps, stream)
        gb = x.numel() * 8 / 1e9
        print("%-22s indexed like x %7.1f us %5.0f GB/s | row-blocked %7.1f us %5.0f GB/s" % (sh, t0 * 1e3, gb / t0 * 1e3, t1 * 1e3, gb / t1 * 1e3),
              flush=True)


def wgrads(args, stream):
    """weight gradient of the step's 3x3 layers: the register-splitting kernel (fp32 operands) against the plane-fed one (both operands'
    row-blocked planes given), us per call incl. the slab-sum launch"""
    for sh in [(2, 256, 64, 64, 256, 3), (2, 128, 128, 128, 128, 3), (2, 512, 32, 32, 512, 3), (2, 256, 128, 128, 256, 3),
               (2, 256, 256, 256, 256, 3), (2, 256, 32, 32, 256, 3), (4, 256, 64, 64, 256, 3)]:
        N, C, Hh, W, Co, k = sh
        g = torch.Generator().manual_seed(1)
        x = cl(torch.randn(N, C, Hh, W, generator=g).relu().cuda())
        dy = cl((torch.randn(N, Co, Hh, W, generator=g) * 1e-3).cuda())
        dw = cl(torch.zeros(Co, C, k, k).cuda())
        db = torch.zeros(Co).cuda()
        with torch.cuda.stream(stream):
            for t in (x, dy):
                t._mmt_amax = H._amax_of(t)
                H.f16_split_pg(t)
        out = []
        for planes in (False, True):
            H.WG_PLANES = planes
            t, _ = timed(lambda: H.conv_wgrad(x, dy, (Co, C, k, k), 1, k // 2, dw, None, db), args.reps, stream)
            out.append(t)
        flop = 2.0 * N * Hh * W * Co * C * k * k
        print("%-28s registers %7.1f us %5.0f TF | planes %7.1f us %5.0f TF" % (sh, out[0] * 1e3, flop / out[0] / 1e9, out[1] * 1e3, flop / out[1] / 1e9),
              flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ablate", action="store_true", help="main-loop ablations (MMT_PG_DBG) of the 64- and 256-row forms on a few shapes")
    ap.add_argument("--split", action="store_true", help="time the plane-split pass in both plane orders")
    ap.add_argument("--wgrad", action="store_true", help="weight gradient of the 3x3 layers: register-splitting vs plane-fed kernel")
    args = ap.parse_args()
    H.lib()
    H.set_conv_precision(3)
    H.set_f16x2(True)
    H.FAST_PLANS = True
    stream = torch.cuda.Stream()
    if args.split:
        return split_passes(args, stream)
    if args.wgrad:
        return wgrads(args, stream)
    if args.ablate:
        return ablate(args, stream)
    print("# us per launch (graph replay of %d calls, best of 5); TF = algorithmic TFLOP/s; peak 833" % args.reps)
    print("%-40s %5s | %8s %6s | %s" % ("shape (N,Cin,H,W,Cout,k,s,p)", "calls", "old us", "TF", "pg variants: rows/ks: us [+split us] TF"))
    tot_old = tot_new = tot_new_split = 0.0
    todo = [SHAPES[i] for i in QUICK] if args.quick else SHAPES
    for sh in todo:
        N, C, Hh, W, Co, k, s, p, calls = sh
        g = torch.Generator().manual_seed(1)
        x = cl(torch.randn(N, C, Hh, W, generator=g).relu().cuda())
        w = cl((torch.randn(Co, C, k, k, generator=g) * (2.0 / (k * k * C)) ** 0.5).cuda())
        b = torch.randn(Co, generator=g).cuda()
        Ho, Wo = (Hh + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        flop = 2.0 * N * Ho * Wo * Co * C * k * k
        with torch.cuda.stream(stream):
            H._amax_of(x)
            xp = H.f16_split_pg(x)
        t_old, mode = timed(lambda: H.conv_forward(x, w, None, b, s, p, relu=True), args.reps, stream)
        t_split, _ = timed(lambda: H.f16_split_pg(x), args.reps, stream)
        rows0, ks0 = H.conv_pg_plan(N, C, Hh, W, Co, k, k, s, p)
        variants = [(rows0, ks0)]
        for rows in (64, 128, 256):
            for ks in (1,):
                tiles = (N * Ho * Wo + rows - 1) // rows * ((Co + 127) // 128)
                if ((rows, ks) not in variants and ks * (256 // rows) <= (C * k * k) // 256 and tiles * ks <= 2048
                        and (ks == 1 or tiles * ks * rows * 512 <= (64 << 20))):
                    variants.append((rows, ks))
        res = []
        for rows, ks in variants:
            try:
                t, _ = timed(lambda: H.conv_forward_pg(x, w, None, b, s, p, relu=True, tile_rows=rows, ksplit=ks, xp=xp), args.reps, stream)
            except RuntimeError as e:
                res.append((rows, ks, None, str(e)[:40]))
                continue
            res.append((rows, ks, t, None))
        t_af, _ = timed(lambda: H.conv_forward_pg(x, w, None, b, s, p, relu=True, xp="fp32"), args.reps, stream)
        ok = [r for r in res if r[2] is not None]
        best = min(ok, key=lambda r: r[2]) if ok else None
        line = "%-40s %5d | %8.1f %6.1f | " % (str(sh[:8]), calls, t_old * 1e3, flop / t_old / 1e9)
        line += " ".join("%d/%d:%.1f" % (r[0], r[1], r[2] * 1e3) if r[2] is not None else "%d/%d:ERR" % (r[0], r[1]) for r in res)
        if best:
            line += "  || plan %d/%d best %d/%d %.1f us %.0f TF, split pass %.1f us, FP32 ROWS (plan, no split pass) %.1f us (%s)" % (
                rows0, ks0, best[0], best[1], best[2] * 1e3, flop / best[2] / 1e9, t_split * 1e3, t_af * 1e3, mode)
            tot_old += calls * t_old
            tot_new += calls * min(best[2], t_old)
            tot_new_split += calls * min(best[2] + t_split, t_old)
        print(line, flush=True)
    print("# per step over these shapes: old %.2f ms, pg (best variant, planes given) %.2f ms, pg + split pass %.2f ms" %
          (tot_old, tot_new, tot_new_split))


if __name__ == "__main__":
    main()
