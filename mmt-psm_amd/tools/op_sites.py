"""Which lines of the package issue the library (ATen) operations of a step.  A TorchDispatchMode on the step thread and on
the teacher thread (backward forced onto the calling thread) charges every ATen call that launches device work to the
innermost Python frame inside the package.  Prints calls per (site, op); view / metadata ops are left out."""
import os, sys, collections, threading, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from torch.utils._python_dispatch import TorchDispatchMode
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, len(sys.argv) > 1 and sys.argv[1] == "irnet", base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
NOKERNEL = ("view", "permute", "slice", "select", "as_strided", "detach", "expand", "unsqueeze", "squeeze", "alias", "empty", "t.",
            "transpose", "_unsafe_view", "reshape", "unbind", "split", "stride", "size", "numel", "is_", "sym_", "_local_scalar",
            "lift_fresh", "narrow", "chunk", "unfold", "set_", "resize_", "record_stream", "_to_copy.default_cpu", "item", "dim",
            "is_pinned", "_pin_memory", "result_type", "can_cast", "_has_compatible")
agg = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(p) or ("." + p) in name for p in NOKERNEL):
            f, site = sys._getframe(1), "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "mmt-psm_amd" in fn and "tools/op_sites" not in fn:
                    site = "%s:%d %s" % (fn.split("mmt-psm_amd/")[-1].replace("maskrcnn_benchmark/", ""), f.f_lineno, f.f_code.co_name)
                    break
                f = f.f_back
            agg[(site, name)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
orig = trainer.teacher.forward_teacher


def wrapped(*a, **k):
    with Sites():
        return orig(*a, **k)


trainer.teacher.forward_teacher = wrapped
il, tg, ul = batch()
with Sites():
    trainer.train_step(1403, il, tg, ul)
torch.cuda.synchronize()
bysite = collections.Counter()
for (s, o), c in agg.items():
    bysite[s] += c
print("library operations of one step (both threads): %d" % sum(agg.values()))
for s, c in bysite.most_common(90):
    ops = sorted(((o, c2) for (s2, o), c2 in agg.items() if s2 == s), key=lambda v: -v[1])[:7]
    print("%5d  %-72s %s" % (c, s[:72], " ".join("%s:%d" % (o.replace(".default", "").replace(".Tensor", ""), n) for o, n in ops)))
