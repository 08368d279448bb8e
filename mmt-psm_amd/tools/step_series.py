"""per-step wall time of the bench step over a few hundred steps next to where the step thread ran (cpu id), the host's
load, and optional pinning -- the bench step runs in two regimes (~36 and ~47 ms) on identical inputs.
env: PIN=local|<cpulist>|'' (sched_setaffinity), SWITCH=<seconds> (sys.setswitchinterval), STEPS"""
import os, sys, time, glob, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
def parse(cl):
    out = set()
    for part in cl.split(","):
        a, _, b = part.partition("-")
        out |= set(range(int(a), int(b or a) + 1))
    return out
import ctypes
getcpu = ctypes.CDLL(None).sched_getcpu
bus = torch.cuda.get_device_properties(0).pci_bus_id
dev = None
for d in glob.glob("/sys/bus/pci/devices/*"):
    if d.lower().endswith(":%02x:00.0" % bus):
        dev = d
local = open(dev + "/local_cpulist").read().strip() if dev else "?"
print("pci", bus, dev, "numa", open(dev + "/numa_node").read().strip() if dev else "?", "local cpus", local,
      "| affinity", len(os.sched_getaffinity(0)), "cpus of", os.cpu_count(), "| loadavg", open("/proc/loadavg").read().strip())
pin = os.environ.get("PIN", "")
if pin:
    cpus = parse(local) if pin == "local" else parse(pin)
    os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
    print("pinned to", len(os.sched_getaffinity(0)), "cpus")
if os.environ.get("SWITCH"):
    sys.setswitchinterval(float(os.environ["SWITCH"]))
# RECIPE_LR=1: the recipe's BASE_LR instead of the bench's frozen one -- the run in which the teacher loses its detections
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=None if os.environ.get("RECIPE_LR") else bench.BENCH_BASE_LR)
N = int(os.environ.get("STEPS", "200"))
rows = []
for i in range(N):
    il, tg, ul = batch()
    trainer.train_step(1400 + i, il, tg, ul)
    torch.cuda.synchronize()
    rows.append((time.perf_counter(), getcpu(), open("/proc/loadavg").read().split()[0], trainer.skipped_pairs))
dts = [(rows[i][0] - rows[i - 1][0]) * 1e3 for i in range(1, N)]
line = ""
for i in range(1, N):
    line += "%d:%.1f%s " % (i, dts[i - 1], "*" if rows[i][3] > rows[i - 1][3] else "")
print("per-step ms (* = consistency branch skipped: the teacher found no box on some unlabeled image):")
print(line)
d = sorted(dts[5:])
print("PIN=%r SWITCH=%r: median %.2f  p10 %.2f  p90 %.2f  mean %.2f  fast(<41) %d of %d" % (
    pin, os.environ.get("SWITCH"), d[len(d) // 2], d[len(d) // 10], d[len(d) * 9 // 10], sum(d) / len(d), sum(1 for x in d if x < 41), len(d)))
