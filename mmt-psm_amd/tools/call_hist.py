"""Which C-ABI entry points the interpreter calls in a training step, and how often (bench.py's `library_calls_per_step` by symbol).
  python call_hist.py [steps]
Counts every call that passes through _hip._check (direct calls) and the recorded launches a plan replays (by symbol as well)."""
import collections
import os
import sys

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H

cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
it0 = cfg.MT.START_MT + cfg.MT.RAMPUP_STEP + 100
for i in range(6):
    il, tg, ul = batch()
    trainer.train_step(it0 + i, il, tg, ul)
torch.cuda.synchronize()

hist = collections.Counter()
_orig = H._check


SITES = set(os.environ.get("MMT_CALL_SITES", "").split(",")) - {""}   # symbols whose callers are listed (MMT_CALL_SITES=mmt_pack_weight,...)
sites = collections.Counter()


def _counting(code, what):
    hist[what] += 1
    if what in SITES:
        import traceback
        fr = [f for f in traceback.extract_stack()[:-1] if "maskrcnn_benchmark" in f.filename and not f.filename.endswith("_hip.py")][-3:]
        sites[(what, " < ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(fr)))] += 1
    return _orig(code, what)


H._check = _counting
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
c0 = H.C_CALLS[0]
for i in range(n):
    il, tg, ul = batch()
    trainer.train_step(it0 + 6 + i, il, tg, ul)
torch.cuda.synchronize()
total = (H.C_CALLS[0] - c0) / n
direct = sum(hist.values()) / n
print("library calls per step: %.1f  (direct %.1f, replayed from launch plans %.1f)" % (total, direct, total - direct))
for k, v in hist.most_common():
    print("  %-40s %7.1f" % (k, v / n))
for (what, where), v in sites.most_common():
    print("  site %-28s %5.1f  %s" % (what, v / n, where))
with H._LP_LOCK:
    plans = list(H._LAUNCH_PLANS.values())
for p in plans:
    names = collections.Counter(getattr(f, "__name__", str(f)) for f, _a in getattr(p, "calls", []))
    segs = getattr(p, "segs", None)
    print("plan: %d recorded launches in %s interpreter calls  %s" % (len(getattr(p, "calls", [])), len(segs) if segs else "?", dict(names)))
