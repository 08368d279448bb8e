"""host-side cost of the teacher forward (it shares the GIL with the step thread): cProfile by own time"""
import cProfile, pstats, io, os, sys, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
il, tg, ul = batch()
tl = [f.to(trainer.device) for f in ul[:trainer.teacher_bs]]
pr = cProfile.Profile()
pr.enable()
with torch.no_grad():
    r = trainer.teacher.forward_teacher(tl)
pr.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 28), ("cumulative", 40)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:]))
