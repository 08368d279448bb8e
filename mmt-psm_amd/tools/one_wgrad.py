"""one weight-gradient shape, timed: N Cin H W Cout k stride pad"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
N, Cin, H, W, Cout, k, s, p = [int(v) for v in sys.argv[1:9]]
x = cl(torch.randn(N, Cin, H, W, device='cuda'))
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
dy = cl(torch.randn(N, Cout, Ho, Wo, device='cuda'))
dw = cl(torch.zeros(Cout, Cin, k, k, device='cuda')); db = torch.zeros(Cout, device='cuda')
for _ in range(3): hip.conv_wgrad(x, dy, (Cout, Cin, k, k), s, p, dw, None, db)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): hip.conv_wgrad(x, dy, (Cout, Cin, k, k), s, p, dw, None, db)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("%s: %.3f ms  %.1f TFLOP/s" % (sys.argv[1:9], ms, 2.0 * N * Ho * Wo * Cout * Cin * k * k / ms / 1e9))
