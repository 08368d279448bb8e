"""one 1x1 forward shape (N x 64 x 64, CIN -> COUT from the environment) launched 30 times: run under
rocprofv3 --kernel-trace --stats to read the tiled kernel's duration as a function of K and of the tile count"""
import os, sys, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
H.lib()
N, CIN, COUT, S = (int(os.environ.get(k, d)) for k, d in (("N", "2"), ("CIN", "1024"), ("COUT", "256"), ("HW", "64")))
g = torch.Generator().manual_seed(0)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(N, CIN, S, S, generator=g).relu_().cuda())
w = cl((torch.randn(COUT, CIN, 1, 1, generator=g) * 0.02).cuda())
sc = torch.ones(COUT).cuda()
for it in range(30):
    y = H.conv_forward(x, w, sc, None, 1, 0, relu=True)
torch.cuda.synchronize()
