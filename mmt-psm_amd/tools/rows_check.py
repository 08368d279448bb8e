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
for N,Cin,H,W,Cout in [(4,64,128,128,256),(2,128,192,192,512),(1,64,257,259,96),(1,128,300,300,200)]:
    torch.manual_seed(0)
    x = cl(torch.randn(N,Cin,H,W,device='cuda')); w = cl(torch.randn(Cout,Cin,1,1,device='cuda')*0.05)
    keep = register(w)
    sc = torch.rand(Cout,device='cuda'); sh = torch.rand(Cout,device='cuda'); res = cl(torch.randn(N,Cout,H,W,device='cuda'))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double()) * sc.double().view(1,-1,1,1) + sh.double().view(1,-1,1,1) + res.double())
    os.environ["MMT_ROWS"] = "1"; y1 = hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1)
    os.environ["MMT_ROWS"] = "0"; y0 = hip.conv_forward(x,w,sc,sh,1,0,relu=True,res=res,res_mode=1)
    print(N,Cin,H,W,Cout, "err rows %.2e generic %.2e  identical %s" % ((y1.double()-ref).abs().max().item()/ref.abs().max().item(), (y0.double()-ref).abs().max().item()/ref.abs().max().item(), torch.equal(y0,y1)))
