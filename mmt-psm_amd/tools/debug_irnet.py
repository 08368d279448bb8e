"""debug helper: full-size IR-Net step, stage by stage with a sync after each; MMT_TRACE=1 prints every library
launch (flushed) before it runs so the last line names a faulting kernel"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from maskrcnn_benchmark import _hip

L = _hip.lib()
if os.environ.get("MMT_TRACE") == "1":
    def wrap(f, name):
        def g(*a):
            desc = ""
            try:
                o = a[0]._obj
                desc = " ".join("%s=%s" % (k, getattr(o, k)) for k, _ in o._fields_ if isinstance(getattr(o, k), int) and k not in ("x", "w", "y"))
            except Exception:
                pass
            print("CALL", name, desc[:300], flush=True)
            r = f(*a)
            torch.cuda.synchronize()
            return r
        return g
    for name in _hip._SIGS:
        setattr(L, name, wrap(getattr(L, name), name))

dev = torch.device("cuda", 0)
cfg, trainer, batch = bench.build(dev, 0, True)
il, targets, ul = batch()
student, teacher = trainer.student, trainer.teacher
def mark(s):
    torch.cuda.synchronize(); print("STAGE ok:", s, flush=True)
mark("build")
out = student(il, targets)
mark("student sup forward " + str({k: float(v.detach()) for k, v in out.items()}))
sum(out.values()).backward()
mark("student sup backward")
il, targets, ul = batch()
trainer.train_step(1400, il, targets, ul)
mark("train_step")
for i in range(12):
    il, targets, ul = batch()
    l = trainer.train_step(1401 + i, il, targets, ul)
    torch.cuda.synchronize()
    print(i, {k: round(float(v.detach()), 4) for k, v in l.items()}, flush=True)
