import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
cases = [(8,64,256,256,256,1),(4,64,256,256,256,1),(8,128,128,128,512,1),(4,128,128,128,512,1),(8,256,64,64,1024,1),(4,256,64,64,1024,1),
         (8,256,256,256,256,1),(4,256,256,256,64,1),(8,256,256,256,128,1),(4,512,128,128,128,1),(4,1024,64,64,256,1),(8,512,32,32,2048,1),(4,2048,32,32,512,1)]
import weakref
class FakeFlat(object):
    pass
def register(w):
    f = FakeFlat(); f.planes = hip.pack_weight(w); f.plane_versions = {w.data_ptr(): w._version}; f.plane_epoch = hip.PLANES_EPOCH + 10**9
    hip.PLANES[w.data_ptr()] = (weakref.ref(f), 0, w.numel())
    return f
for N,Cin,H,W,Cout,k in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    keep = register(w)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda'); res = cl(torch.randn(N,Cout,H,W,device='cuda'))
    for _ in range(3): y = hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    it=20; e0.record()
    for _ in range(it): y = hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/it
    fl = 2.0*N*H*W*Cout*Cin
    by = 4.0*N*H*W*(Cin+2*Cout)
    print("N%d %4d->%4d @%3d  %7.3f ms %6.1f TF/s %5.2f TB/s" % (N,Cin,Cout,H,ms,fl/ms/1e9,by/ms/1e9))
