"""which calls make H.nhwc() copy (layout conversions on the hot path): shape + caller, one step"""
import collections, os, sys, traceback, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(2):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
log = collections.Counter()
orig = H.nhwc
def spy(x):
    if x.dim() == 4 and not x.permute(0, 2, 3, 1).is_contiguous():
        fr = traceback.extract_stack(limit=4)
        log[(tuple(x.shape), tuple(x.stride()), " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[:-1][::-1]))] += 1
    return orig(x)
H.nhwc = spy
import maskrcnn_benchmark.layers.fused as Fz
il, tg, ul = batch(); trainer.train_step(1402, il, tg, ul)
torch.cuda.synchronize()
for (sh, st, who), n in sorted(log.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel())[:25]:
    print("%2d x %-24s strides %-30s %.1f MB  %s" % (n, sh, st, 4e-6 * torch.Size(sh).numel(), who))
