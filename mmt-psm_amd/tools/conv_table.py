"""Every convolution call of one train_step by shape: count, time, algorithmic TFLOP/s, and the time a perfect kernel would
need (max of the MFMA time at 2500 / products-per-multiply TFLOP/s -- 3 products on the default fp16 split, 6 on the bf16 split --
and fp32 HBM bytes at 5 TB/s).  Single stream (MMT_OVERLAP_TEACHER=0)
so the event brackets hold one kernel's own time.  Sorted by the time above that bound -- the list of what is left."""
import os, sys, collections, torch
os.environ.setdefault("MMT_OVERLAP_TEACHER", "0")
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H
if os.environ.get("MMT_LIB"):   # A/B of two builds of the library on one box (tools only)
    H.LIB_PATH = os.path.abspath(os.environ["MMT_LIB"])
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(4):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
H.PROFILE, H.PROFILE_ALL = [], True
STEPS = 3
for i in range(STEPS):
    il, tg, ul = batch(); trainer.train_step(1410 + i, il, tg, ul)
torch.cuda.synchronize()
rec, H.PROFILE = H.PROFILE, None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rec:
    fl, e0, e1, key = r[0], r[1], r[2], r[3]
    a = agg[key]
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] = fl
    a[3] += r[6] if len(r) > 6 else 0.0     # bytes of the fused epilogue's operands (residual, ReLU mask, dropout multiplier)
NPROD = 3 if (H.F16X2 and H.get_conv_precision() == 3) else bench.PRODUCTS[H.get_conv_precision()]   # matrix products per multiply
rows = []
for key, (c, ms, fl, eb) in agg.items():
    kind, N, Hh, W, Cin, Cout, KH, stride, ostride = key
    Ho, Wo = (Hh + stride - 1) // stride, (W + stride - 1) // stride
    if kind == "wgrad":
        byts = 4.0 * N * (Hh * W * Cin + Ho * Wo * Cout) + 4.0 * Cin * Cout * KH * KH
    else:
        byts = 4.0 * N * (Hh * W * Cin + Ho * Wo * Cout * (ostride if ostride else 1)) + 6.0 * Cin * Cout * KH * KH
    byts += eb / c
    t_m, t_h = fl / (2500e12 / NPROD) * 1e3, byts / 5e12 * 1e3
    per = ms / c
    rows.append((c / STEPS * (per - max(t_m, t_h)), key, c / STEPS, per, fl / per / 1e9, t_m, t_h))
rows.sort(key=lambda r: -r[0])
tot = sum(r[2] * r[3] for r in rows)
print("total conv time per step %.2f ms; bound %.2f ms (%d matrix products per multiply)" % (tot, sum(r[2] * max(r[5], r[6]) for r in rows), NPROD))
print("%-46s %6s %8s %8s %8s %8s %8s" % ("shape (kind,N,H,W,Cin,Cout,K,s,os)", "n/step", "ms each", "TFLOP/s", "t_mfma", "t_hbm", "excess"))
cat = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for ex, key, c, per, tf, t_m, t_h in rows:
    k = ("wgrad " if key[0] == "wgrad" else "fwd/dgrad ") + ("fc" if key[2] == 1 else "%dx%d" % (key[6], key[6])) + (" N=%d" % key[1] if key[2] > 1 and key[1] <= 8 else "")
    cat[k][0] += c; cat[k][1] += c * per; cat[k][2] += c * max(t_m, t_h)
for k, (c, ms, b) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print("  %-24s %6.1f calls %7.2f ms  bound %6.2f ms  (%.2f)" % (k, c, ms, b, b / ms))
for ex, key, c, per, tf, t_m, t_h in rows[:int(os.environ.get("ROWS", "60"))]:
    print("%-46s %6.1f %8.3f %8.1f %8.3f %8.3f %8.2f" % (str(key).replace(" ", ""), c, per, tf, t_m, t_h, ex))
