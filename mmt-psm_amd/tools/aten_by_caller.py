"""ATen (library) kernel launches of one train_step by the innermost line of this repo that issued them: count and GPU time.
The list of what is left to fuse (f-2)."""
import os, sys, collections, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    il, tg, ul = batch(); trainer.train_step(1403, il, tg, ul)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
own = [0, 0.0]
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    lib = [k for k in ks if k.name.startswith("void at::") or k.name.startswith("at::") or "rocprim" in k.name or "Cijk" in k.name or "hipcub" in k.name]
    if not lib:
        own[0] += len(ks); own[1] += sum(k.duration for k in ks)
        continue
    st, q = [], e
    while q is not None and not st:   # the kernel hangs on the innermost op; the python stack is on an outer one
        st = [f for f in (q.stack or []) if "mmt-psm_amd" in f or "/bench.py" in f]
        q = q.cpu_parent
    who = st[0].split("mmt-psm_amd/")[-1].replace("maskrcnn_benchmark/", "")[:100] if st else "(autograd engine / no python frame)"
    a = agg[who]
    a[0] += len(lib); a[1] += sum(k.duration for k in lib); a[2][e.name] += len(lib)
tot = sum(v[0] for v in agg.values())
print("library launches: %d, %.2f ms; own-kernel launches %d, %.2f ms" % (tot, sum(v[1] for v in agg.values()) / 1e3, own[0], own[1] / 1e3))
by_file = collections.Counter()
for who, (c, t, ops) in agg.items():
    by_file[who.split("(")[0].split(".py")[0]] += c
print("by file:", by_file.most_common(12))
for who, (c, t, ops) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("ROWS", "70"))]:
    print("%4d  %7.1f us  %-100s %s" % (c, t, who, dict(ops.most_common(4))))
