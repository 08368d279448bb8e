"""host-side profile (cProfile) of one mean-teacher step: where the Python/launch time goes"""
import cProfile, pstats, sys, os, io, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, len(sys.argv) > 1 and sys.argv[1] == "irnet", base_lr=bench.BENCH_BASE_LR)
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
if os.environ.get('MMT_BW_INLINE'):
    torch.autograd.set_multithreading_enabled(False)   # backward on the calling thread: visible to cProfile
pr = cProfile.Profile()
il, tg, ul = batch()
if os.environ.get('MMT_PROFILE_TEACHER'):   # the profile of the TEACHER thread instead (cProfile is per thread)
    tp = cProfile.Profile()
    orig = trainer.teacher.forward_teacher
    def wrapped(*a, **k):
        tp.enable()
        try:
            return orig(*a, **k)
        finally:
            tp.disable()
    trainer.teacher.forward_teacher = wrapped
    trainer.train_step(1403, il, tg, ul)
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(tp, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])
    sys.exit(0)
pr.enable()
trainer.train_step(1403, il, tg, ul)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:14000])
