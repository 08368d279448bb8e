import sys, torch, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
cases = [
 ("fpn_layer1 3x3 256@256^2 N2", 2,256,256,256,256,3,1,1),
 ("rpn 3x3 256@128^2 N2", 2,256,128,128,256,3,1,1),
 ("l1 1x1 64->256 @256^2 N2", 2,64,256,256,256,1,1,0),
 ("l2 3x3 128@128^2 N2", 2,128,128,128,128,3,1,1),
 ("l3 3x3 256@64^2 N2", 2,256,64,64,256,3,1,1),
 ("l3 1x1 256->1024 @64^2 N2", 2,256,64,64,1024,1,1,0),
 ("l4 3x3 512@32^2 N2", 2,512,32,32,512,3,1,1),
 ("l4 1x1 512->2048 @32^2 N2", 2,512,32,32,2048,1,1,0),
 ("l4 3x3 512@32^2 N8", 8,512,32,32,512,3,1,1),
 ("fpn_layer1 N8", 8,256,256,256,256,3,1,1),
 ("stem 7x7 N2", 2,4,1024,1024,64,7,2,3),
 ("fc6 R1024", 1024,12544,1,1,1024,1,1,0),
 ("mask 3x3 256@14^2 P256", 256,256,14,14,256,3,1,1),
]
for name,N,Cin,H,W,Cout,k,s,p in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    for _ in range(3): y = hip.conv_forward(x,w,sc,sh,s,p,relu=True)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    it=10; e0.record()
    for _ in range(it): y = hip.conv_forward(x,w,sc,sh,s,p,relu=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/it
    Ho=(H+2*p-k)//s+1; Wo=(W+2*p-k)//s+1
    fl = 2.0*N*Ho*Wo*Cout*Cin*k*k
    print("%-32s %8.3f ms  %7.1f TFLOP/s" % (name, ms, fl/ms/1e9))
    # wgrad
    if Cin%4==0 and Cin>4:
        dy = torch.randn_like(y); dw = torch.zeros_like(w)
        for _ in range(2): hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw)
        torch.cuda.synchronize(); e0.record()
        for _ in range(it): hip.conv_wgrad(x,dy,(Cout,Cin,k,k),s,p,dw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/it
        print("%-32s %8.3f ms  %7.1f TFLOP/s  (wgrad)" % ("", ms, fl/ms/1e9))
