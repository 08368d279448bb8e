"""the fp16-split weight-gradient kernel on a few shapes of the step, 20 launches each (run under rocprofv3 --kernel-trace --stats)"""
import os, sys, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
H.lib()
g = torch.Generator().manual_seed(0)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
N, C, S, Co, k = (int(os.environ.get(a, d)) for a, d in (("N", "2"), ("CIN", "256"), ("HW", "64"), ("COUT", "256"), ("K", "3")))
x = cl(torch.randn(N, C, S, S, generator=g).relu_().cuda()); dy = cl((torch.randn(N, Co, S, S, generator=g) * 1e-3).cuda())
dw = cl(torch.zeros(Co, C, k, k).cuda())
for it in range(20):
    H.conv_wgrad(x, dy, (Co, C, k, k), 1, k // 2, dw)
torch.cuda.synchronize()
