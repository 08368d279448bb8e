import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hip.set_conv_precision(mode)
for name,N,Cin,H,W,Cout,k,s,p in [("fpn_layer1 N8", 8,256,256,256,256,3,1,1), ("l3 1x1 1024->256 N8", 8,1024,64,64,256,1,1,0)]:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    import weakref
    class F(object): pass
    f = F(); f.planes = hip.pack_weight(w); f.plane_versions = {w.data_ptr(): w._version}; f.plane_epoch = hip.PLANES_EPOCH + 10**9
    hip.PLANES[w.data_ptr()] = (weakref.ref(f), 0, w.numel())
    for _ in range(3): y = hip.conv_forward(x,w,None,None,s,p)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    it=10; e0.record()
    for _ in range(it): y = hip.conv_forward(x,w,None,None,s,p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/it
    fl = 2.0*N*H*W*Cout*Cin*k*k
    print("DBG=%s mode %d %-24s %7.3f ms %6.1f TF" % (os.environ.get("MMT_DBG","0"), mode, name, ms, fl/ms/1e9), flush=True)
