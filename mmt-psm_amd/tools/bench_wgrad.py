"""weight gradient: error vs fp64 and TFLOP/s per arithmetic mode"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
checks = [("3x3 s1 64@24^2", 2,64,24,24,96,3,1,1), ("1x1 s2 128->160", 3,128,30,30,160,1,2,0), ("3x3 odd 48->64", 1,48,17,23,64,3,1,1),
          ("linear 1024x256->512", 1000,256,1,1,512,1,1,0), ("3x3 Cout=36 Cin=20", 2,20,12,12,36,3,1,1), ("mask 3x3 14^2", 20,256,14,14,256,3,1,1)]
for name,N,Cin,H,W,Cout,k,s,p in checks:
    torch.manual_seed(0)
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda'))
    Ho=(H+2*p-k)//s+1; Wo=(W+2*p-k)//s+1
    dy = cl(torch.randn(N,Cout,Ho,Wo,device='cuda'))
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=s, padding=p)
    for mode in (0,3,2,1):
        hip.set_conv_precision(mode)
        dw = torch.zeros_like(w); db = torch.zeros(Cout, device='cuda')
        hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw,None,db)
        err = (dw.double()-ref).abs().max().item()/ref.abs().max().item()
        eb = (db.double()-dy.double().sum((0,2,3))).abs().max().item()
        print("%-24s mode %d  dw max err %.3e  dbias err %.2e" % (name, mode, err, eb))
cases = [
 ("fpn_layer1 3x3 256@256^2 N2", 2,256,256,256,256,3,1,1),
 ("l1 1x1 64->256 @256^2 N4", 4,64,256,256,256,1,1,0),
 ("l2 3x3 128@128^2 N4", 4,128,128,128,128,3,1,1),
 ("l3 3x3 256@64^2 N4", 4,256,64,64,256,3,1,1),
 ("l3 1x1 256->1024 @64^2 N4", 4,256,64,64,1024,1,1,0),
 ("l4 3x3 512@32^2 N4", 4,512,32,32,512,3,1,1),
 ("l4 1x1 2048->512 @32^2 N4", 4,2048,32,32,512,1,1,0),
 ("fc6 R1024", 1024,12544,1,1,1024,1,1,0),
 ("mask 3x3 256@14^2 P128", 128,256,14,14,256,3,1,1),
]
for name,N,Cin,H,W,Cout,k,s,p in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda'))
    Ho=(H+2*p-k)//s+1; Wo=(W+2*p-k)//s+1
    dy = cl(torch.randn(N,Cout,Ho,Wo,device='cuda'))
    fl = 2.0*N*Ho*Wo*Cout*Cin*k*k
    line = "%-28s" % name
    d0 = None
    for mode in (0,3,2,1):
        hip.set_conv_precision(mode)
        dw = cl(torch.zeros(Cout,Cin,k,k,device='cuda'))
        for _ in range(2): hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        it=10; e0.record()
        for _ in range(it): hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/it
        dw.zero_(); hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw)
        if d0 is None: d0 = dw.clone()
        err = (dw-d0).abs().max().item()/d0.abs().max().item()
        line += " | m%d %6.3f ms %6.1f TF err %.1e" % (mode, ms, fl/ms/1e9, err)
    print(line, flush=True)
hip.set_conv_precision(0)
