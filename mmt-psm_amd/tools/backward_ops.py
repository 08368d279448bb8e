"""The library (ATen) operations the autograd engine itself issues during the two backward passes of a step -- gradient
accumulation of multi-consumer tensors, backward nodes of tensor arithmetic used in the forward -- by operation and operand shapes.
They run on the step's critical chain (round 5: every small launch there costs its duration plus a launch gap).
    python mmt-psm_amd/tools/backward_ops.py"""
import os, sys, collections, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.utils._python_dispatch import TorchDispatchMode
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, False, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
NOKERNEL = ("view", "permute", "slice", "select", "as_strided", "detach", "expand", "unsqueeze", "squeeze", "alias", "empty", "t.",
            "transpose", "_unsafe_view", "reshape", "unbind", "split", "stride", "size", "numel", "is_", "sym_", "_local_scalar",
            "lift_fresh", "narrow", "chunk", "unfold", "set_", "resize_", "record_stream", "item", "dim", "is_pinned", "_pin_memory",
            "result_type", "can_cast", "_has_compatible")
agg = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(p) or ("." + p) in name for p in NOKERNEL):
            f, site = sys._getframe(1), "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "mmt-psm_amd" in fn and "tools/" not in fn:
                    site = "%s:%s" % (os.path.basename(fn), f.f_code.co_name)
                    break
                f = f.f_back
            if site.startswith("MTtrainer.py:_backward_roots") or site.startswith("MTtrainer.py:train_step"):
                shapes = tuple(tuple(a.shape) if torch.is_tensor(a) else (type(a).__name__ if not isinstance(a, (int, float)) else a) for a in args[:3])
                agg[(name.replace(".default", "").replace(".Tensor", ""), shapes)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
il, tg, ul = batch()
with Sites():
    trainer.train_step(1403, il, tg, ul)
torch.cuda.synchronize()
print("operations issued by the autograd engine / the step function itself in one step: %d" % sum(agg.values()))
for (op, shapes), c in agg.most_common(80):
    print("%4d  %-22s %s" % (c, op, shapes))
