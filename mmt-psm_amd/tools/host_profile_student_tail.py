"""cProfile of the ONE stretch of a step where the device waits for the host (profiles/r03_host_device_phases.txt): the main
thread's student.forward_student after the teacher's join -- MGD, the student's box head on the teacher's proposals, PSM"""
import cProfile, pstats, sys, os, io, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(5):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
pr = cProfile.Profile()
orig = trainer.student.forward_student
times = []
def wrapped(*a, **k):
    t0 = time.perf_counter()
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
        times.append(time.perf_counter() - t0)
trainer.student.forward_student = wrapped
for i in range(5):
    il, tg, ul = batch(); trainer.train_step(1410 + i, il, tg, ul)
torch.cuda.synchronize()
print("forward_student host time per call (ms, under cProfile):", ["%.2f" % (t * 1e3) for t in times])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:12000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
