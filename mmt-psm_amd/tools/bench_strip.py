"""tap-strip 3x3 kernel (planes in, 8 waves, 256x128 tiles) vs the in-register-split kernel: time, error vs fp64-free
reference (the old kernel), split-pass cost"""
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
cases = [("fpn 3x3 256@256^2 N8", 8,256,256,256,256), ("fpn 3x3 256@256^2 N2", 2,256,256,256,256),
         ("rpn 3x3 256@128^2 N8", 8,256,128,128,256), ("rpn 3x3 256@128^2 N2", 2,256,128,128,256),
         ("l1 3x3 64@256^2 N8", 8,64,256,256,64), ("l2 3x3 128@128^2 N8", 8,128,128,128,128),
         ("l3 3x3 256@64^2 N8", 8,256,64,64,256), ("p4 3x3 256@64^2 N8 (res)", 8,256,64,64,256),
         ("p4 3x3 256@64^2 N2 splitK", 2,256,64,64,256), ("l2 3x3 128@128^2 N2 splitK", 2,128,128,128,128),
         ("p4 3x3 256@64^2 N4 splitK", 4,256,64,64,256)]
for name,N,Cin,H,W,Cout in cases:
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,3,3,device='cuda')*0.05)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda')
    res = cl(torch.randn(N,Cout,H,W,device='cuda')) if "res" in name else None
    kw = dict(relu=True, res=res, res_mode=1 if res is not None else 0)
    fl = 2.0*N*H*W*Cout*Cin*9
    os.environ["MMT_STRIP"] = "0"
    y0 = hip.conv_forward(x,w,sc,sh,1,1,**kw)
    t0 = timeit(lambda: hip.conv_forward(x,w,sc,sh,1,1,**kw))
    os.environ["MMT_STRIP"] = "1"
    xp = hip.split_planes(x)
    y1 = hip.conv_forward(x,w,sc,sh,1,1,x_planes=xp,**kw)
    t1 = timeit(lambda: hip.conv_forward(x,w,sc,sh,1,1,x_planes=xp,**kw))
    ts = timeit(lambda: hip.split_planes(x, xp))
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), None, 1, 1) * sc.double().view(1,-1,1,1) + sh.double().view(1,-1,1,1)
    if res is not None: ref = ref + res[:1].double()
    ref = torch.relu(ref)
    e0 = ((y0[:1].double()-ref).abs().max()/ref.abs().max()).item(); e1 = ((y1[:1].double()-ref).abs().max()/ref.abs().max()).item()
    print("%-28s old %7.3f ms %6.1f TF | strip %7.3f ms %6.1f TF | split %6.3f ms | incl. %6.1f TF | err vs fp64 old %.2e new %.2e | max|new-old| %.2e" % (
        name, t0, fl/t0/1e9, t1, fl/t1/1e9, ts, fl/(t1+ts)/1e9, e0, e1, (y0-y1).abs().max().item()), flush=True)
