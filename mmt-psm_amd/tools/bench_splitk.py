"""split-K on / off (MMT_SPLITK read at library load: run twice) for the few-tile, long-K shapes"""
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
for N,Cin,H,W,Cout,k in [(2,256,64,64,256,3),(2,512,32,32,512,3),(2,256,32,32,256,3),(2,256,16,16,256,3),(2,1024,64,64,256,1),(2,2048,32,32,512,1),(2,512,32,32,2048,1),
                         (1024,12544,1,1,1024,1),(2,128,128,128,128,3),(4,256,64,64,256,3),(27,256,14,14,256,3)]:
    torch.manual_seed(0)
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.02)
    keep = register(w)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), None, 1, k//2) * sc.double().view(1,-1,1,1) + sh.double().view(1,-1,1,1))
    y = hip.conv_forward(x,w,sc,sh,1,k//2,relu=True)
    y2 = hip.conv_forward(x,w,sc,sh,1,k//2,relu=True)
    err = (y.double()-ref).abs().max().item()/ref.abs().max().item()
    for _ in range(3): hip.conv_forward(x,w,sc,sh,1,k//2,relu=True)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.conv_forward(x,w,sc,sh,1,k//2,relu=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print("N%d %4d->%4d @%3d k%d  %.3f ms %6.1f TF  err %.2e  repeatable %s" % (N,Cin,Cout,H,k,ms,2.0*N*H*W*Cout*Cin*k*k/ms/1e9,err,torch.equal(y,y2)))
