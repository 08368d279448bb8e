"""One convolution shape launched repeatedly (for `rocprofv3 --pmc ...`): python pmc_one.py N H W Cin Cout k [reps]"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maskrcnn_benchmark import _hip as H  # noqa: E402

N, Hh, W, Cin, Cout, k = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
H.set_f16x2(True)
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Cin, Hh, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
for _ in range(reps):
    y = H.conv_forward(x, w, sc, sh, 1, k // 2, relu=True)
torch.cuda.synchronize()
