"""the teacher's forward on its own: wall time, kernel time, idle gaps (it is the critical path of the step's forward)"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
def run():
    il, tg, ul = batch()
    tl = [f.to(trainer.device) for f in ul[:trainer.teacher_bs]]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        r = trainer.teacher.forward_teacher(tl)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for _ in range(3):
    print("teacher alone: host %.1f ms, done %.1f ms" % run())
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
ev = prof.events()
dev = sorted((e.time_range.start, e.time_range.end) for e in ev if str(e.device_type).endswith("CUDA") and e.time_range.end > e.time_range.start)
busy = sum(e - s for s, e in dev)
gaps = [(dev[i + 1][0] - max(d[1] for d in dev[:i + 1][-3:])) for i in range(len(dev) - 1)]
big = [g for g in gaps if g > 15]
print("kernels %d, busy %.1f ms, span %.1f ms, gaps > 15 us: %d totalling %.1f ms, > 100 us: %d totalling %.1f ms" % (
    len(dev), busy / 1e3, (dev[-1][1] - dev[0][0]) / 1e3, len(big), sum(big) / 1e3, len([g for g in gaps if g > 100]), sum(g for g in gaps if g > 100) / 1e3))
