import os, sys, torch, weakref
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
class FakeFlat(object): pass
def register(w):
    f = FakeFlat(); f.planes = hip.pack_weight(w); f.plane_versions = {w.data_ptr(): w._version}; f.plane_epoch = hip.PLANES_EPOCH + 10**9
    hip.PLANES[w.data_ptr()] = (weakref.ref(f), 0, w.numel())
    return f
def tm(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
for N,Cin,H,W,Cout in [(8,64,256,256,256),(8,128,128,128,512),(8,256,64,64,1024),(8,256,256,256,64)]:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,1,1,device='cuda')*0.05)
    keep = register(w)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda'); res = cl(torch.randn(N,Cout,H,W,device='cuda'))
    y = torch.empty_like(res)
    M = N*H*W
    a = tm(lambda: hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1))
    b = tm(lambda: hip.conv_forward(x,w,sc,sh,1,0,relu=True))
    c = tm(lambda: y.copy_(res))
    d = tm(lambda: torch.add(res, res, out=y))
    print("%d->%d M=%d: res %.3f ms (%.2f TB/s) | no res %.3f ms (%.2f TB/s) | copy %.3f ms (%.2f TB/s) | add(2 reads) %.3f" % (
        Cin,Cout,M,a,4.0*M*(Cin+2*Cout)/a/1e9,b,4.0*M*(Cin+Cout)/b/1e9,c,8.0*M*Cout/c/1e9,d))
