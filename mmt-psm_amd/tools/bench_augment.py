"""input augmentation: device pipeline vs the Pillow path of the reference, 1000x1000 -> 800x800, K = 3 views"""
import os, sys, time, random
import numpy as np
import torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark.config import make_default_cfg
from maskrcnn_benchmark.data.transforms import build_transforms, DeviceImage
from maskrcnn_benchmark.data.transforms.transforms import augment_views
cfg = make_default_cfg()
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (1000, 1000, 3), dtype=np.uint8) for _ in range(8)]
dev = [DeviceImage(torch.from_numpy(i).cuda()) for i in imgs]   # decoded pixels resident in HBM
tr = build_transforms(cfg, True, "no_label")
def one(d):
    base, _ = tr[0](d, None)
    return augment_views(base, tr[1], 3, size_divisible=32)
for d in dev: one(d)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for rep in range(10):
    for d in dev:
        v = one(d); n += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
px = 800 * 800
byt = 1000 * 1000 * 3 + 800 * 1000 * 3 * 2 + px * 3 + 3 * (2 * px * 3 + px * 12)   # resample passes + 3 x (luma read, view read, fp32 write)
print("device: %.3f ms per unlabeled sample (resize + flip + 3 views, host draws included) = %.0f samples/s; "
      "algorithmic %.1f MB/sample -> %.2f TB/s" % (dt * 1e3, 1 / dt, byt / 1e6, byt / dt / 1e12))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for d in dev: one(d)
e1.record(); torch.cuda.synchronize()
print("device, GPU time only: %.3f ms per sample" % (e0.elapsed_time(e1) / len(dev)))
# the reference's CPU path for the same work (Pillow, one core)
from oracle import transforms as OT
t0 = time.perf_counter()
for i in imgs[:3]:
    OT.pipeline(i, "no_label", 3, cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, random, np.random, restated=False)
dtc = (time.perf_counter() - t0) / 3
print("Pillow path (reference, 1 core): %.1f ms per sample = %.1f samples/s" % (dtc * 1e3, 1 / dtc))
