"""What would the step cost if a family of launches were free?  Timing only (results are wrong by construction): the bench's step with
a group of launches skipped, to see which device work the step is actually waiting for before building a faster kernel for it.
  python whatif.py            -> baseline | no weight gradients | no plane-split passes | no optimiser / EMA / re-pack
Every arm: 8 warm-up + 30 timed steps, mean and median of the per-step device time."""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import fused

cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
it0 = cfg.MT.START_MT + cfg.MT.RAMPUP_STEP + 100


def run(tag, n=30, warm=8):
    for i in range(warm):
        il, tg, ul = batch()
        trainer.train_step(it0 + i, il, tg, ul)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    for i in range(n):
        marks[i].record()
        il, tg, ul = batch()
        trainer.train_step(it0 + warm + i, il, tg, ul)
    marks[n].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    print("%-44s %.2f ms/step  (median %.2f)" % (tag, dt, statistics.median(per)), flush=True)


run("baseline")
# ---- no weight gradients at all (conv / fc / deconv): what the side stream's 7 ms of kernel time cost the step
_cw, _wg = H.conv_wgrad, fused._wgrad
H.conv_wgrad = lambda *a, **k: None


def _no_wgrad(x, g, w, stride, pad, rowscale=None, with_bias=False, dst_w=None, dst_b=None):
    fused._touch(dst_w, dst_b if with_bias else None)
    return (None if dst_w is not None else torch.zeros_like(w)), (None if (dst_b is not None or not with_bias) else torch.zeros((w.shape[0],), device=w.device))


fused._wgrad = _no_wgrad
run("no weight gradients")
H.conv_wgrad, fused._wgrad = _cw, _wg
# ---- weight gradients on the step stream (no side stream)
fused._WG_ON = False
run("weight gradients on the step stream")
fused._WG_ON = True
# ---- no optimiser step / EMA / plane re-pack
_step, _upd = trainer.optimizer.step, trainer.update_teacher
trainer.optimizer.step = lambda *a, **k: None
trainer.update_teacher = lambda *a, **k: None
run("no SGD / EMA / weight re-pack")
trainer.optimizer.step, trainer.update_teacher = _step, _upd
# ---- teacher on the step stream (no overlap): the sum of the two chains
trainer.overlap_teacher = False
run("teacher on the step stream")
trainer.overlap_teacher = True
run("baseline again")
