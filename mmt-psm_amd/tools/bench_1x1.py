"""isolated timing of the 1x1 bottleneck / lateral shapes of the step (forward with BN + ReLU (+ residual))"""
import sys, torch, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
cases = [(8,256,64,64,1024,True),(8,1024,64,64,256,False),(2,256,64,64,1024,True),(2,1024,64,64,256,False),
         (8,128,128,128,512,True),(8,512,128,128,128,False),(2,128,128,128,512,True),(2,512,128,128,128,False),
         (8,512,32,32,2048,True),(8,2048,32,32,512,False),(2,512,32,32,2048,True),(2,2048,32,32,512,False),
         (8,256,256,256,256,False),(2,256,256,256,256,False),(8,256,256,256,64,False),(8,64,256,256,256,True)]
for N,Cin,H,W,Cout,resid in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,1,1,device='cuda')*0.05)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    res = cl(torch.randn(N,Cout,H,W,device='cuda')) if resid else None
    f = lambda: hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1 if resid else 0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    fl = 2.0*N*H*W*Cout*Cin
    by = 4.0*N*H*W*(Cin+Cout*(2 if resid else 1))
    a = hip.ConvArgs()
    print("N%d %4d->%4d @%3d^2 res=%d  %7.3f ms %6.1f TF %6.0f GB/s (min bytes)" % (N,Cin,Cout,H,resid,ms,fl/ms/1e9,by/ms/1e6), flush=True)
