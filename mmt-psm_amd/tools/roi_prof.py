import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/mmt-psm_amd")
import bench
from maskrcnn_benchmark import _hip
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(2):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
orig = _hip.roi_align_backward
rec = []
def wrapped(grad, shapes, scales, rois, lv, ph, pw, sr):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(grad, shapes, scales, rois, lv, ph, pw, sr); e1.record()
    rec.append((rois.shape[0], ph, pw, torch.bincount(lv.long(), minlength=4).tolist(), e0, e1))
    return r
_hip.roi_align_backward = wrapped
import maskrcnn_benchmark.layers.fused as fused
il, tg, ul = batch(); trainer.train_step(1402, il, tg, ul)
torch.cuda.synchronize()
for k, ph, pw, lvs, e0, e1 in rec:
    print("K=%d %dx%d levels=%s  %.3f ms" % (k, ph, pw, lvs, e0.elapsed_time(e1)))
