"""one conv shape in a loop (for rocprofv3 --pmc): N Cin H W Cout k [res]"""
import os, sys, torch, weakref
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
class FakeFlat(object): pass
N, Cin, H, W, Cout, k = [int(v) for v in sys.argv[1:7]]
use_res = len(sys.argv) > 7 and sys.argv[7] == "res"
x = cl(torch.randn(N, Cin, H, W, device='cuda')); w = cl(torch.randn(Cout, Cin, k, k, device='cuda') * 0.05)
f = FakeFlat(); f.planes = hip.pack_weight(w); f.plane_versions = {w.data_ptr(): w._version}; f.plane_epoch = hip.PLANES_EPOCH + 10**9
hip.PLANES[w.data_ptr()] = (weakref.ref(f), 0, w.numel())
sc = torch.rand(Cout, device='cuda'); sh = torch.rand(Cout, device='cuda')
res = cl(torch.randn(N, Cout, H, W, device='cuda')) if use_res else None
for _ in range(10):
    y = hip.conv_forward(x, w, sc, sh, 1, k // 2, relu=True, res=res, res_mode=1 if use_res else 0)
torch.cuda.synchronize()
