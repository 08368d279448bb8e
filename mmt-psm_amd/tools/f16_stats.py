"""fp16 two-term split (default arithmetic of mode 3): how many launches of a step take it, which tensors still need a reduction
pass of their own (nobody recorded their maximum), and the weight-gradient kernel on it vs the 3-term bf16 split"""
import os, sys, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark import _hip as H
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
H.lib()
g = torch.Generator().manual_seed(0)
for N, C, S, Co in ((2, 256, 256, 256), (2, 256, 64, 256), (2, 128, 128, 128), (8, 256, 256, 256)):
    x = cl(torch.randn(N, C, S, S, generator=g).relu_().cuda()); dy = cl((torch.randn(N, Co, S, S, generator=g) * 1e-4).cuda())
    dw = cl(torch.zeros(Co, C, 3, 3, device="cuda"))
    fl = 2.0 * N * S * S * Co * C * 9
    H.set_f16x2(False)
    t3 = timeit(lambda: H.conv_wgrad(x, dy, (Co, C, 3, 3), 1, 1, dw))
    H.set_f16x2(True)
    x._mmt_amax = (x.abs().max().reshape(1), x._version); dy._mmt_amax = (dy.abs().max().reshape(1), dy._version)
    th = timeit(lambda: H.conv_wgrad(x, dy, (Co, C, 3, 3), 1, 1, dw))
    H.set_f16x2(False)
    print("wgrad 3x3 %d x %d @ %d^2 -> %d: bf16 x3 %.3f ms %.0f TF | fp16 x2 %.3f ms %.0f TF" % (N, C, S, Co, t3, fl / t3 / 1e9, th, fl / th / 1e9))
H.set_f16x2(True)
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
for k in H.F16_STATS: H.F16_STATS[k] = 0
H.AMAX_LOG = []
il, tg, ul = batch(); trainer.train_step(1403, il, tg, ul)
torch.cuda.synchronize()
print("launches of one step on the fp16 split:", H.F16_STATS)
import collections
c = collections.Counter(H.AMAX_LOG)
print("reduction passes of one step (tensor shape, call chain):")
for (shape, chain), n in sorted(c.items(), key=lambda kv: -kv[1] * 1e9 - (kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3] if len(kv[0][0]) == 4 else 0)):
    print("  %2d x %-22s %s" % (n, shape, chain))
