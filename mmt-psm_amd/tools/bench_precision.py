"""conv forward: error vs fp64 and TFLOP/s for each arithmetic mode (0 fp32 MFMA, 3/2/1 split-bf16)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
import ctypes
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
class FakeFlat(object):
    pass
def register(w):
    import weakref
    f = FakeFlat(); f.planes = hip.pack_weight(w); f.plane_versions = {w.data_ptr(): w._version}; f.plane_epoch = hip.PLANES_EPOCH + 10**9
    hip.PLANES[w.data_ptr()] = (weakref.ref(f), 0, w.numel())
    return f
cases = [
 ("fpn_layer1 3x3 256@256^2 N2", 2,256,256,256,256,3,1,1),
 ("fpn_layer1 N8", 8,256,256,256,256,3,1,1),
 ("l1 1x1 64->256 @256^2 N4", 4,64,256,256,256,1,1,0),
 ("l2 3x3 128@128^2 N4", 4,128,128,128,128,3,1,1),
 ("l3 3x3 256@64^2 N4", 4,256,64,64,256,3,1,1),
 ("l3 1x1 256->1024 @64^2 N4", 4,256,64,64,1024,1,1,0),
 ("l4 3x3 512@32^2 N8", 8,512,32,32,512,3,1,1),
 ("l4 1x1 2048->512 @32^2 N8", 8,2048,32,32,512,1,1,0),
 ("fc6 R1024", 1024,12544,1,1,1024,1,1,0),
 ("mask 3x3 256@14^2 P256", 256,256,14,14,256,3,1,1),
]
for name,N,Cin,H,W,Cout,k,s,p in [("check 3x3 s1 64@24^2", 2,64,24,24,96,3,1,1), ("check 1x1 s2", 3,128,30,30,160,1,2,0), ("check 3x3 odd", 1,48,17,23,64,3,1,1),
                                  ("check 3x3 big", 2,256,64,64,256,3,1,1), ("check 1x1 K=16", 2,16,40,40,64,1,1,0), ("check 1x1 K=32 128x64", 8,32,64,64,64,1,1,0)]:
    torch.manual_seed(0)
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, s, p)
    for glds in (0, 1):
        keep = register(w) if glds else None
        for mode in (0,3,2,1):
            hip.set_conv_precision(mode)
            y = hip.conv_forward(x,w,None,None,s,p)
            err = (y.double()-ref).abs().max().item()/ref.abs().max().item()
            print("%-24s planes=%d mode %d  max err %.3e" % (name, glds, mode, err))
        hip.PLANES.clear()
for name,N,Cin,H,W,Cout,k,s,p in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    keep = register(w)
    Ho=(H+2*p-k)//s+1; Wo=(W+2*p-k)//s+1
    fl = 2.0*N*Ho*Wo*Cout*Cin*k*k
    line = "%-28s" % name
    y0 = None
    for mode in (0,3,2,1):
        hip.set_conv_precision(mode)
        for _ in range(3): y = hip.conv_forward(x,w,sc,sh,s,p,relu=True)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        it=10; e0.record()
        for _ in range(it): y = hip.conv_forward(x,w,sc,sh,s,p,relu=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/it
        if y0 is None: y0 = y
        err = (y-y0).abs().max().item()/y0.abs().max().item()
        line += " | m%d %6.3f ms %6.1f TF err %.1e" % (mode, ms, fl/ms/1e9, err)
    print(line, flush=True)
    hip.PLANES.clear()
hip.set_conv_precision(0)
