"""one shape, a few launches of the 3x3 conv in the chosen kernel (MMT_STRIP=0/1) -- target for rocprofv3 --pmc"""
import sys, torch, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as hip
hip.lib()
def cl(x): return x.contiguous(memory_format=torch.channels_last)
N,Cin,H,W,Cout = 8,256,256,256,256
x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,3,3,device='cuda')*0.05)
xp = hip.split_planes(x) if os.environ.get("MMT_STRIP","1") != "0" else None
for _ in range(4):
    y = hip.conv_forward(x,w,None,None,1,1,x_planes=xp)
torch.cuda.synchronize()
