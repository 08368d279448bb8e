"""is the teacher thread slowed by the step thread's Python work (GIL) or by its GPU work?  teacher job + (a) idle main thread,
(b) main thread spinning in pure Python, (c) main thread launching GPU work only (one big matmul loop, little Python)"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
def probe(kind):
    il, tg, ul = batch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    job = trainer._start_teacher(ul)
    if kind == "spin":
        t = time.perf_counter()
        x = 0
        while time.perf_counter() - t < 0.022:
            x += 1
    elif kind == "gpu":
        for _ in range(12):  # ~22 ms of GEMM at ~600 TF/s... adjust: 8192^3*2 = 1.1 TF each
            torch.mm(a, b)
    job["thread"].join()
    t1 = time.perf_counter()
    trainer.t_stream.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for kind in ("idle", "spin", "gpu", "idle", "spin", "gpu"):
    print("%-5s teacher thread joined after %.1f ms, all GPU work done after %.1f ms" % ((kind,) + probe(kind)))
