"""device time of one backbone pass: launch by launch vs replayed as a captured hipGraph (engine/graphs.py)"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
from maskrcnn_benchmark.engine.graphs import BackboneGraph
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for n, model, train in ((8, trainer.teacher, False), (2, trainer.student, True)):
    x = torch.randn(n, 3, 1024, 1024, device="cuda")
    def eager():
        with torch.set_grad_enabled(train):
            o = model.backbone(x)
        if train:
            req = [t for t in o if t.requires_grad]
            torch.autograd.backward(req, [torch.zeros_like(t) for t in req])
    for _ in range(3): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eager()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    g = BackboneGraph(model.backbone, x, train, trainer.flat_s if train else None)
    def replay():
        o = g(x)
        if train:
            req = [t for t in o if t.requires_grad]
            torch.autograd.backward(req, [torch.zeros_like(t) for t in req])
    for _ in range(3): replay()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    for _ in range(5): replay()
    t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print("N=%d train=%s: eager host %.2f ms, device-complete %.2f ms | graph host %.2f ms, device-complete %.2f ms" % (
        n, train, (t1 - t0) / 5e-3, (t2 - t0) / 5e-3, (t4 - t3) / 5e-3, (t5 - t3) / 5e-3))
