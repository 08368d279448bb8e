"""timing experiments on the all-planes kernel (library built with -DMMT_PP_EXPERIMENTS): what each part of the step costs"""
import sys, torch, os, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from maskrcnn_benchmark import _hip as hip
    hip.lib()
    def cl(x): return x.contiguous(memory_format=torch.channels_last)
    N,Cin,H,W,Cout,k,s,p = 8,256,256,256,256,3,1,1
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,k,k,device='cuda')*0.05)
    xp = hip.split_planes(x)
    f = lambda: hip.conv_forward(x,w,None,None,s,p,x_planes=xp)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print("DBG=%s  %.3f ms  %.1f TF" % (os.environ.get("MMT_PP_DBG","0"), ms, 2.0*N*H*W*Cout*Cin*k*k/ms/1e9), flush=True)
else:
    for d in ["0","1","2","3","4","7"]:
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, MMT_PP_DBG=d))
