"""Host-side timeline of one overlapped train_step: when each thread enters / leaves the big modules and the backward calls
(perf_counter only, no device syncs added), next to the step's wall time."""
import os, sys, time, threading, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
LOG = []
def stamp(what):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()   # on the calling thread's current stream: the device reaches this point when everything queued before is done
    LOG.append((time.perf_counter(), threading.current_thread().name, what, ev))
def hook(model, tag):
    for name in ("backbone", "rpn", "box_heads", "mask_heads", "hint_adaptor"):
        m = getattr(model, name, None)
        if m is None: continue
        m.register_forward_pre_hook(lambda mod, inp, n=name: stamp("%s.%s >" % (tag, n)))
        m.register_forward_hook(lambda mod, inp, out, n=name: stamp("%s.%s <" % (tag, n)))
hook(trainer.student, "S"); hook(trainer.teacher, "T")
ob = torch.Tensor.backward
def bw(self, *a, **k):
    stamp("backward >"); r = ob(self, *a, **k); stamp("backward <"); return r
torch.Tensor.backward = bw
oab = torch.autograd.backward
for nm in ("forward_source", "forward_unlabel", "update_teacher"):
    f = getattr(trainer, nm)
    def w(*a, _f=f, _n=nm, **k):
        stamp(_n + " >"); r = _f(*a, **k); stamp(_n + " <"); return r
    setattr(trainer, nm, w)
ost = trainer.optimizer.step
def st(*a, **k):
    stamp("optimizer >"); r = ost(*a, **k); stamp("optimizer <"); return r
trainer.optimizer.step = st
for i in range(4):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
for rep in range(2):
    il, tg, ul = batch()
    torch.cuda.synchronize()
    del LOG[:]
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    t0 = time.perf_counter()
    trainer.train_step(1410 + rep, il, tg, ul)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step: host returns at %.2f ms, device done at %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print("  host issue time | device time (when the stream got there) | thread | phase")
    for t, th, what, ev in LOG:
        print("  %7.2f ms  %7.2f ms  %-12s %s" % ((t - t0) * 1e3, e0.elapsed_time(ev), th, what))
