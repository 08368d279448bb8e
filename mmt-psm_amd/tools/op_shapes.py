"""GPU time of one torch op family by input shape and python caller (argv[1] = op name, e.g. aten::fill_)"""
import os, sys, collections, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    il, tg, ul = batch(); trainer.train_step(1403, il, tg, ul)
    torch.cuda.synchronize()
ops = sys.argv[1].split(",")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ops and getattr(e, "kernels", None):
        st = [f for f in (e.stack or []) if "mmt-psm_amd" in f or "bench.py" in f]
        who = st[0].split("/")[-1][:60] if st else "?"
        agg[(e.name, str(e.input_shapes)[:60], who)][0] += 1
        agg[(e.name, str(e.input_shapes)[:60], who)][1] += sum(k.duration for k in e.kernels)
for (n, sh, who), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%3d x %8.1f us  %-14s %-60s %s" % (c, t, n, sh, who))
