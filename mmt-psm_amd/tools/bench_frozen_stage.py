"""Device time (graph replay) of the frozen-stage / special-shape kernels of csrc/conv_stem.hip against the launches they replace:
layer1's 3x3 (MMT_C64), the RPN predictors (MMT_THIN), the fused stem (backbone._STEM_FUSED)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "mmt-psm_amd"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "mmt-psm_amd", "tools"))
from maskrcnn_benchmark import _hip as H
import bench_pg
H.lib(); H.set_conv_precision(3); H.set_f16x2(True)
st = torch.cuda.Stream()
for N in (8, 4):
    x = torch.randn(N, 64, 256, 256).relu().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3) * 0.06).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(64).cuda()
    with torch.cuda.stream(st):
        H._amax_of(x)
    for e in ("1", "0"):
        os.environ["MMT_C64"] = e
        t, mode = bench_pg.timed(lambda: H.conv_forward(x, w, None, b, 1, 1, relu=True), 10, st)
        print("N=%d C64=%s: %.1f us (%s)" % (N, e, t * 1e3, mode))
from maskrcnn_benchmark.config import make_default_cfg
from maskrcnn_benchmark.modeling.backbone import backbone as B
m = B.StemWithFixedBatchNorm(make_default_cfg()).cuda()
for N in (8, 4):
    x = (torch.randn(N, 3, 1024, 1024) * 60).cuda()
    with torch.cuda.stream(st):
        H._amax_of(x)
    for e in (True, False):
        B._STEM_FUSED[0] = e
        with torch.no_grad():
            t, mode = bench_pg.timed(lambda: m(x), 5, st)
        print("stem N=%d fused=%s: %.1f us (%s)" % (N, e, t * 1e3, mode))
