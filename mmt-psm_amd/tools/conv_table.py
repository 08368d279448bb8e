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
NPROD = 3 if (H.F16X2 and H.get_conv_precision() == 3) else bench.PRODUCTS[H.get_conv_precision()]   # matrix products per multiply
table, fam = bench.conv_family(rec, STEPS, NPROD)   # (the figure bench.py prints as roofline.family comes from the same function)
rows = sorted(((c * (per - max(t_m, t_h)), key, c, per, fl / per / 1e9, t_m, t_h) for key, c, per, fl, t_m, t_h in table), key=lambda r: -r[0])
print("total conv time per step %.2f ms; bound %.2f ms = %.3f of it (%d matrix products per multiply; %.0f TB/s)" % (
    fam["time_ms"], fam["bound_ms"], fam["frac"], NPROD, bench.HBM_ACHIEVABLE_TBS))
print("%-46s %6s %8s %8s %8s %8s %8s" % ("shape (kind,N,H,W,Cin,Cout,K,s,os)", "n/step", "ms each", "TFLOP/s", "t_mfma", "t_hbm", "excess"))
for k, v in fam["groups"].items():
    print("  %-24s %6.1f calls %7.2f ms  bound %6.2f ms  (%.2f)" % (k, v["launches"], v["time_ms"], v["bound_ms"], v["frac"]))
for ex, key, c, per, tf, t_m, t_h in rows[:int(os.environ.get("ROWS", "60"))]:
    print("%-46s %6.1f %8.3f %8.1f %8.3f %8.3f %8.2f" % (str(key).replace(" ", ""), c, per, tf, t_m, t_h, ex))
