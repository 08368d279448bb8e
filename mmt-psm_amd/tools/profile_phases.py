"""Tuning aid: wall/GPU time of the phases of one mean-teacher step + kernel-launch counts (torch profiler)."""
import os
import sys
import time

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.cuda.set_device(0)
cfg, tr, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
it0 = 1400


def phases(i):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    cpu = [0.0] * 8
    il, tg, ul = batch()
    ev[0].record(); cpu[0] = time.perf_counter()
    xs, xu = il.tensors, ul[-1].tensors
    pyr = tr.student.backbone(torch.cat([xs, xu], 0))
    n = xs.shape[0]
    fs, fu = tuple(l[:n] for l in pyr), [tuple(l[n:] for l in pyr)]
    ev[1].record(); cpu[1] = time.perf_counter()
    ld = tr.forward_source(il, tg, fs)
    ev[2].record(); cpu[2] = time.perf_counter()
    with torch.no_grad():
        tres = tr.teacher.forward_teacher([f for f in ul[:2]])
    ev[3].record(); cpu[3] = time.perf_counter()
    ld.update(tr.student.forward_student(ul[-1:], tres, features=fu))
    ev[4].record(); cpu[4] = time.perf_counter()
    losses = sum(tr.weight_sum_loss(ld, it0 + i).values())
    tr.optimizer.zero_grad()
    losses.backward()
    ev[5].record(); cpu[5] = time.perf_counter()
    tr.optimizer.step()
    tr.update_teacher(100)
    ev[6].record(); cpu[6] = time.perf_counter()
    torch.cuda.synchronize()
    names = ["student backbone (N=4)", "sup heads+losses", "teacher (all)", "student unsup heads+losses", "backward",
             "SGD+EMA"]
    return [(names[k], ev[k].elapsed_time(ev[k + 1]), (cpu[k + 1] - cpu[k]) * 1e3) for k in range(6)]


for i in range(3):
    r = phases(i)
print("%-30s %10s %10s" % ("phase", "GPU ms", "host ms"))
for n, g, c in r:
    print("%-30s %10.2f %10.2f" % (n, g, c))
print("total GPU %.1f ms" % sum(g for _, g, _ in r))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    phases(5)
ka = prof.key_averages()
cnt = sum(e.count for e in ka if e.device_type == torch.autograd.DeviceType.CUDA) if hasattr(torch.autograd, "DeviceType") else -1
print("kernel launches in one step:", cnt)
rows = sorted([e for e in ka if e.self_device_time_total > 0 and not e.key.startswith("void") and "kernel" not in e.key.lower()],
              key=lambda e: -e.self_device_time_total)[:40]
for e in rows:
    print("%-44s calls %5d  self GPU %8.3f ms  self cpu %7.2f ms" % (e.key[:44], e.count, e.self_device_time_total / 1e3,
                                                                   e.self_cpu_time_total / 1e3))
