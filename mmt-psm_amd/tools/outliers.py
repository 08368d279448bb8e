"""how often a bench step is an outlier (> 1.15 x the median), with a switch of the library on and off in ONE process order-alternated
across processes: tells a box's own jitter from a code path that is slow now and then (the exact path of a range guard)"""
import os, sys, time, statistics, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
N = int(os.environ.get("STEPS", "150"))
ts = []
for i in range(N + 10):
    il, tg, ul = batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.train_step(1400 + i, il, tg, ul)
    torch.cuda.synchronize()
    if i >= 10:
        ts.append((time.perf_counter() - t0) * 1e3)
med = statistics.median(ts)
out = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 1.15 * med]
print("%s: median %.2f mean %.2f  outliers %d of %d: %s" % (os.environ.get("TAG", ""), med, sum(ts) / len(ts), len(out), len(ts), out[:20]), flush=True)
