"""Which launches of one step take a plane-split pass in front (mmt_split_planes_f16_rb), and who PRODUCED the tensor they split
(round 6: the list of producer epilogues that have to write row-blocked planes themselves).  One line per (consumer, producer)."""
import collections
import os
import sys

import torch

os.environ["MMT_LAUNCH_PLANS"] = "0"   # (a replayed pass issues no Python calls: nothing to log)
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H

LOG = []
_conv_forward, _split = H.conv_forward, H.f16_split_pg


def _chain(depth, skip=2):
    f = sys._getframe(skip)
    out = []
    while f is not None and len(out) < depth:
        out.append("%s:%d" % (f.f_code.co_name, f.f_lineno))
        f = f.f_back
    return " < ".join(out)


def conv_forward(x, w, *a, **k):
    y = _conv_forward(x, w, *a, **k)
    src = w if w is not None else k.get("f16_src", (None,))[0]
    shp = tuple(src.shape) if src is not None else None
    y._mmt_src = ("conv", "dgrad" if w is None else "fwd", shp, tuple(x.shape), k.get("res_mode", 0) if k.get("res") is not None else 0,
                  k.get("mask") is not None, k.get("stride", a[2] if len(a) > 2 else 1))
    return y


def f16_split_pg(x):
    r = _split(x)
    if not r[3]:   # (lag 0: a split pass ran; 1: the producer's epilogue had written the planes)
        LOG.append((tuple(x.shape), getattr(x, "_mmt_src", None), _chain(6)))
    return r


H.conv_forward = conv_forward
H.f16_split_pg = f16_split_pg
from maskrcnn_benchmark.layers import fused as F   # noqa: E402  (module-level references `H.conv_forward`: patched above)

cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(4):
    il, tg, ul = batch()
    trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
LOG.clear()
il, tg, ul = batch()
trainer.train_step(1403, il, tg, ul)
torch.cuda.synchronize()
c = collections.Counter(LOG)
tot = 0
print("split passes of one step: %d" % len(LOG))
for (shape, src, chain), n in sorted(c.items(), key=lambda kv: -kv[1] * (kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3])):
    mb = n * shape[0] * shape[1] * shape[2] * shape[3] * 8 / 1e6
    tot += mb
    print("%2d x %-20s %7.1f MB  producer %-90s consumer %s" % (n, shape, mb, src, chain))
print("total %.0f MB of split traffic per step" % tot)
