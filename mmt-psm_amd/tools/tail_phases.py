"""GPU time of the phases after the teacher is back (the serial tail of the step): events on the main stream"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
ev = {}
def mark(k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev[k] = e
orig_fs = trainer.student.forward_student
def fs(*a, **k):
    mark("unsup_fwd_start"); r = orig_fs(*a, **k); mark("unsup_fwd_end"); return r
trainer.student.forward_student = fs
orig_step = trainer.optimizer.step
def ostep(*a, **k):
    mark("backward_end"); r = orig_step(*a, **k); mark("sgd_end"); return r
trainer.optimizer.step = ostep
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
for i in range(3):
    il, tg, ul = batch()
    torch.cuda.synchronize(); mark("start")
    trainer.train_step(1410 + i, il, tg, ul)
    mark("end"); torch.cuda.synchronize()
    t = lambda a, b: ev[a].elapsed_time(ev[b])
    print("step %.1f | to unsup fwd start %.1f | unsup heads fwd %.1f | unsup backward %.1f | sgd %.1f | ema+pack %.1f" % (
        t("start", "end"), t("start", "unsup_fwd_start"), t("unsup_fwd_start", "unsup_fwd_end"), t("unsup_fwd_end", "backward_end"),
        t("backward_end", "sgd_end"), t("sgd_end", "end")))
