"""Tuning aid: per-shape time of every conv / wgrad launch in one mean-teacher step (events on the launch stream)."""
import collections
import os
import sys

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from maskrcnn_benchmark import _hip  # noqa: E402

torch.cuda.set_device(0)
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
it0 = 1400
for i in range(2):
    il, tg, ul = batch()
    trainer.train_step(it0 + i, il, tg, ul)
torch.cuda.synchronize()
_hip.PROFILE, _hip.PROFILE_ALL = [], True
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
il, tg, ul = batch()
trainer.train_step(it0 + 2, il, tg, ul)
e1.record()
torch.cuda.synchronize()
prof, _hip.PROFILE = _hip.PROFILE, None
agg = collections.OrderedDict()
for fl, a, b, key in (q[:4] for q in prof):
    t = a.elapsed_time(b)
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += t
    d[2] += fl
tot = sum(v[1] for v in agg.values())
print("step %.1f ms; conv+wgrad launches %d, %.1f ms, %.1f TFLOP/s overall" % (
    e0.elapsed_time(e1), len(prof), tot, sum(v[2] for v in agg.values()) / tot / 1e9))
bykind = collections.defaultdict(lambda: [0.0, 0.0])
for k, v in agg.items():
    bykind[k[0]][0] += v[1]
    bykind[k[0]][1] += v[2]
for k, v in bykind.items():
    print("  %-6s %7.2f ms  %6.1f TFLOP/s" % (k, v[0], v[1] / v[0] / 1e9))
def min_bytes(key):
    # input + output once (fp32); residual / mask operands not counted
    kind, N, H, W, Cin, Cout, k, s, _ = key
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    return 4.0 * N * (H * W * Cin + Ho * Wo * Cout)


print("%-6s %3s %5s %5s %5s %5s %2s %1s %1s | %4s %8s %8s %8s %7s" % (
    "kind", "N", "H", "W", "Cin", "Cout", "k", "s", "o", "n", "ms", "TF/s", "GB/s", "floor%"))
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    by = min_bytes(key) * v[0]
    floor = max(v[2] / 416.7e9, by / 6.0e9)  # ms at the split-bf16 peak / at 6 TB/s
    print("%-6s %3d %5d %5d %5d %5d %2d %1d %1d | %4d %8.3f %8.1f %8.0f %7.0f" % (
        key + (v[0], v[1], v[2] / v[1] / 1e9, by / v[1] / 1e6, 100 * floor / v[1])))
