"""kernel-level timing of the student's few-tile convolutions in isolation (run under rocprofv3 --kernel-trace --stats):
3x3 256->256 at 2 x 64 x 64 (strip split-K + finish + plane split), 1x1 1024->256 and 256->1024 at 2 x 64 x 64, their weight gradients"""
import os, sys, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
H.lib()
g = torch.Generator().manual_seed(0)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
N, S = int(os.environ.get("N", "2")), 64
x = cl(torch.randn(N, 256, S, S, generator=g).relu_().cuda())
w3 = cl((torch.randn(256, 256, 3, 3, generator=g) * 0.02).cuda())
x1 = cl(torch.randn(N, 1024, S, S, generator=g).relu_().cuda())
w1 = cl((torch.randn(256, 1024, 1, 1, generator=g) * 0.02).cuda())
w2 = cl((torch.randn(1024, 256, 1, 1, generator=g) * 0.02).cuda())
sc = torch.ones(256).cuda(); sc2 = torch.ones(1024).cuda()
mask = cl(torch.randn(N, 256, S, S, generator=g).cuda())
dw3 = cl(torch.zeros(256, 256, 3, 3).cuda()); dw1 = cl(torch.zeros(256, 1024, 1, 1).cuda())
for it in range(30):
    y = H.conv_forward(x, w3, sc, None, 1, 1, relu=True, mask=mask)          # 3x3 (strip, split-K)
    y1 = H.conv_forward(x1, w1, sc, None, 1, 0, relu=True)                   # reducing 1x1 (tiled)
    y2 = H.conv_forward(y1, w2, sc2, None, 1, 0, relu=True, res=x1, res_mode=1)   # expanding 1x1 (rows)
    H.conv_wgrad(x, y, (256, 256, 3, 3), 1, 1, dw3)
    H.conv_wgrad(x1, y1, (256, 1024, 1, 1), 1, 0, dw1)
torch.cuda.synchronize()
