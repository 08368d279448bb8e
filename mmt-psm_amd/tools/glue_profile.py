"""which host-level blocks own the small ("glue") kernels of a step: kernel count and GPU time per scope"""
import os, sys, collections, functools, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity, record_function
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
import maskrcnn_benchmark.modeling.rpn.rpn as R
import maskrcnn_benchmark.modeling.roi_heads.box_head.box_head as B
import maskrcnn_benchmark.modeling.roi_heads.mask_head.mask_head as M
import maskrcnn_benchmark.modeling.detector.generalized_rcnn as G
import maskrcnn_benchmark.modeling.rpn.anchor_generator as A
import maskrcnn_benchmark.engine.MTtrainer as T


def wrap(obj, name, tag):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        with record_function("SCOPE:" + tag):
            return f(*a, **k)
    setattr(obj, name, g)


wrap(R.RPNLossComputation, "__call__", "rpn.loss")
wrap(R.RPNPostProcessor, "compute_candidates", "rpn.candidates")
wrap(R.RPNPostProcessor, "select", "rpn.select")
wrap(A.AnchorGenerator, "forward", "anchors")
wrap(B.FastRCNNLossComputation, "subsample", "box.subsample")
wrap(B.FastRCNNLossComputation, "__call__", "box.loss")
wrap(B.FastRCNNLossComputation, "evaluatePSM", "box.psm")
wrap(B.PostProcessor, "forward", "box.postprocess")
wrap(M.MaskRCNNLossComputation, "__call__", "mask.loss+targets")
wrap(M.MaskPostProcessor, "forward", "mask.postprocess")
wrap(G.GeneralizedRCNN, "get_fg_feature_loss", "mgd")
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    il, tg, ul = batch(); trainer.train_step(1403, il, tg, ul)
    torch.cuda.synchronize()
ev = prof.events()
scopes = [(e.time_range.start, e.time_range.end, e.name[6:]) for e in ev if e.name.startswith("SCOPE:")]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = [0, 0.0]
for e in ev:
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        continue
for e in ev:
    ks = e.kernels if hasattr(e, "kernels") else []
    if not ks:
        continue
    t0 = e.time_range.start
    owner = "(other)"
    best = None
    for s0, s1, n in scopes:
        if s0 <= t0 <= s1 and (best is None or s1 - s0 < best):
            owner, best = n, s1 - s0
    for k in ks:
        agg[owner][0] += 1
        agg[owner][1] += k.duration
        tot[0] += 1
        tot[1] += k.duration
other = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    ks = e.kernels if hasattr(e, "kernels") else []
    if not ks:
        continue
    t0 = e.time_range.start
    if any(s0 <= t0 <= s1 for s0, s1, _ in scopes):
        continue
    for k in ks:
        other[e.name][0] += 1
        other[e.name][1] += k.duration
print("-- launches outside the scopes, by op")
for n, (c, t) in sorted(other.items(), key=lambda kv: -kv[1][0])[:28]:
    print("   %-40s kernels %5d   GPU %7.2f ms" % (n[:40], c, t / 1e3))
print("total kernels %d, %.2f ms" % (tot[0], tot[1] / 1e3))
inscope = collections.defaultdict(lambda: collections.Counter())
for e in ev:
    ks = e.kernels if hasattr(e, "kernels") else []
    if not ks:
        continue
    t0 = e.time_range.start
    owner, best = None, None
    for s0, s1, n in scopes:
        if s0 <= t0 <= s1 and (best is None or s1 - s0 < best):
            owner, best = n, s1 - s0
    if owner:
        inscope[owner][e.name] += len(ks)
for n in os.environ.get("SCOPES", "").split(","):
    if n in inscope:
        print("-- ops in", n, " ".join("%s:%d" % (k.replace("aten::", ""), v) for k, v in inscope[n].most_common(40)))
# GPU idle time in front of each kernel, charged to the scope (or op, outside scopes) that launched the kernel
kl = []
launch = []  # (cpu start, owner) per launched kernel, in launch order (single stream: run with MMT_OVERLAP_TEACHER=0)
for e in ev:
    ks = e.kernels if hasattr(e, "kernels") else []
    if not ks:
        continue
    t0 = e.time_range.start
    owner, best = None, None
    for s0, s1, n in scopes:
        if s0 <= t0 <= s1 and (best is None or s1 - s0 < best):
            owner, best = n, s1 - s0
    for k in ks:
        launch.append((t0, owner or ("op:" + e.name)))
launch.sort(key=lambda v: v[0])
dev = sorted((e.time_range.start, e.time_range.end - e.time_range.start) for e in ev
             if str(e.device_type).endswith("CUDA") and e.time_range.end > e.time_range.start)
print("launched %d, device events %d" % (len(launch), len(dev)))
for (st, du), (_, ow) in zip(dev, launch):
    kl.append((st, du, ow))
kl.sort()
idle = collections.defaultdict(lambda: [0, 0.0])
end = kl[0][0] + kl[0][1]
for st, du, ow in kl[1:]:
    if st - end > 3:
        idle[ow][0] += 1
        idle[ow][1] += st - end
    end = max(end, st + du)
print("-- GPU idle in front of kernels, by launching scope / op (total %.2f ms)" % (sum(v[1] for v in idle.values()) / 1e3))
for n, (c, t) in sorted(idle.items(), key=lambda kv: -kv[1][1])[:22]:
    print("   %-40s gaps %5d   idle %7.2f ms" % (n[:40], c, t / 1e3))
host = collections.defaultdict(lambda: [0, 0.0])
for s0, s1, n in scopes:
    host[n][0] += 1
    host[n][1] += s1 - s0
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-22s kernels %5d   GPU %7.2f ms   host: %2d calls %7.2f ms" % (n, c, t / 1e3, host[n][0], host[n][1] / 1e3))
