import os, sys, torch
sys.path.insert(0, "/root/repo/mmt-psm_amd")
from maskrcnn_benchmark import _hip as H
H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(1)
for name, N, Cin, Hh, W, Cout in (("n32 h32 w64", 32, 128, 32, 64, 128), ("n8 h32 w64", 8, 128, 32, 64, 128), ("n32 64x64 c128", 32, 128, 64, 64, 128), ("n8 c128 64x64", 8, 128, 64, 64, 128), ("tw64 splitk", 2, 256, 64, 64, 256), ("tw64", 8, 256, 64, 64, 256), ("tw128 splitk", 2, 128, 128, 128, 128),
                                   ("tw128 N2", 2, 256, 128, 128, 256), ("tw64 cout512", 4, 512, 64, 64, 512), ("cout 384", 2, 256, 128, 128, 384)):
    x = cl(torch.randn(N, Cin, Hh, W, generator=g).relu_().cuda()); w = cl((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).cuda())
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda(); sh = (torch.randn(Cout, generator=g) * 0.1).cuda()
    res = cl(torch.randn(N, Cout, Hh, W, generator=g).cuda()); mask = cl(torch.randn(N, Cout, Hh, W, generator=g).cuda())
    for kw in (dict(), dict(relu=True, res=res, res_mode=1), dict(mask=mask, mask_scale=1.5)):
        H.set_f16x2(False)
        y3 = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
        H.set_f16x2(True)
        yh = H.conv_forward(x, w, sc, sh, 1, 1, **kw)
        H.set_f16x2(False)
        d = (y3 - yh).abs().max().item() / y3.abs().max().item()
        print("%-14s %-28s ksplit? rel diff %.2e" % (name, sorted(kw), d))
