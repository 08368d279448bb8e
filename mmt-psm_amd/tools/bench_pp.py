"""all-planes 3x3 kernel vs the in-register-split kernel: time, bit-identity, split pass cost"""
import sys, torch, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
cases = [("fpn 3x3 256@256^2 N8", 8,256,256,256,256,3,1,1), ("fpn 3x3 256@256^2 N2", 2,256,256,256,256,3,1,1),
         ("rpn 3x3 256@128^2 N8", 8,256,128,128,256,3,1,1), ("l2 3x3 128@128^2 N8", 8,128,128,128,128,3,1,1),
         ("l3 3x3 256@64^2 N8", 8,256,64,64,256,3,1,1), ("l4 3x3 512@32^2 N8", 8,512,32,32,512,3,1,1),
         ("l3 1x1 1024->256 @64^2 N8", 8,1024,64,64,256,1,1,0), ("mask 3x3 256@14^2 P256", 256,256,14,14,256,3,1,1)]
for name,N,Cin,H,W,Cout,k,s,p in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    fl = 2.0*N*H*W*Cout*Cin*k*k
    y0 = hip.conv_forward(x,w,sc,sh,s,p,relu=True)
    xp = hip.split_planes(x)
    y1 = hip.conv_forward(x,w,sc,sh,s,p,relu=True,x_planes=xp)
    same = torch.equal(y0,y1)
    t0 = timeit(lambda: hip.conv_forward(x,w,sc,sh,s,p,relu=True))
    t1 = timeit(lambda: hip.conv_forward(x,w,sc,sh,s,p,relu=True,x_planes=xp))
    ts = timeit(lambda: hip.split_planes(x, xp))
    print("%-28s glds %7.3f ms %6.1f TF | planes %7.3f ms %6.1f TF | split %6.3f ms | incl. split %6.1f TF | bit-identical %s maxdiff %.2e" % (
        name, t0, fl/t0/1e9, t1, fl/t1/1e9, ts, fl/(t1+ts)/1e9, same, (y0-y1).abs().max().item()))
