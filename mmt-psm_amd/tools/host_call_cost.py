"""host cost per call of the Python entry points of the hot path (no device sync inside the timed loops; the kernels are tiny)"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import fused
H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(2, 64, 16, 16, device="cuda"))
w = cl(torch.randn(64, 64, 1, 1, device="cuda"))
w3 = cl(torch.randn(64, 64, 3, 3, device="cuda"))
s, b = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
def t(name, f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-44s %6.1f us per call" % (name, (t1 - t0) / n * 1e6))
t("torch.empty_like", lambda: torch.empty_like(x))
t("aten add (x + x)", lambda: x + x)
t("H.nhwc(x)", lambda: H.nhwc(x))
t("H.conv_forward 1x1 (BN+ReLU)", lambda: H.conv_forward(x, w, s, b, 1, 0, relu=True))
t("H.conv_forward 3x3 + residual", lambda: H.conv_forward(x, w3, s, b, 1, 1, relu=True, res=x, res_mode=1))
dw = torch.zeros_like(w)
t("H.conv_wgrad 1x1", lambda: H.conv_wgrad(x, x, (64, 64, 1, 1), 1, 0, dw))
xr = x.clone().requires_grad_(True)
wr = [w.clone().requires_grad_(True), w3.clone().requires_grad_(True), cl(torch.randn(256, 64, 1, 1, device="cuda")).requires_grad_(True),
      cl(torch.randn(256, 64, 1, 1, device="cuda")).requires_grad_(True)]
bn = (s, b, s, b, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"), torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"))
def blk():
    return fused.BottleneckFn.apply(xr, wr[0], wr[1], wr[2], wr[3], bn, 1)
t("BottleneckFn forward (4 convs)", blk, 500)
def blk_fb():
    o = blk(); o.backward(o.detach())
t("BottleneckFn forward + backward (13 launches)", blk_fb, 300)
if os.environ.get("MMT_HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): blk_fb()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
