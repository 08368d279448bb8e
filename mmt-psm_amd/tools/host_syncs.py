"""Where the step thread (and the teacher thread) block on the device in one overlapped train_step: every runtime call that
waits (synchronize / memcpy / event wait on the host) longer than 15 us, with its offset from the step start, its thread and
the innermost repo frame; plus host launch counts per ms."""
import os, sys, collections, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(4):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
il, tg, ul = batch()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_step(1404, il, tg, ul)
    torch.cuda.synchronize()
evs = [e for e in prof.events()]
t0 = min(e.time_range.start for e in evs)
rt = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and (e.name.startswith("hip") or e.name.startswith("cuda"))]
names = collections.Counter(e.name for e in rt)
print("runtime calls:", dict(names.most_common(12)))
threads = sorted({e.thread for e in rt})
print("threads:", threads)
blk = [e for e in rt if ("ynchronize" in e.name or "Memcpy" in e.name or "EventQuery" in e.name) and e.time_range.elapsed_us() > 15]
tot = collections.defaultdict(float)
for e in sorted(blk, key=lambda e: e.time_range.start):
    st = [f for f in (e.stack or []) if "mmt-psm_amd" in f or "bench.py" in f]
    who = st[0].split("mmt-psm_amd/")[-1][:90] if st else "?"
    tot[e.thread] += e.time_range.elapsed_us()
    print("t=%7.2f ms  thread %d  %-28s %8.1f us  %s" % ((e.time_range.start - t0) / 1e3, threads.index(e.thread), e.name, e.time_range.elapsed_us(), who))
print("blocked per thread (ms):", {threads.index(k): round(v / 1e3, 2) for k, v in tot.items()})
# launches per ms per thread
lp = collections.defaultdict(collections.Counter)
for e in rt:
    if "Launch" in e.name:
        lp[threads.index(e.thread)][int((e.time_range.start - t0) // 1000)] += 1
for th, c in lp.items():
    print("thread", th, "launches/ms:", " ".join("%d" % c.get(ms, 0) for ms in range(max(c) + 1)))
