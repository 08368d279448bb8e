"""wall-clock split of one step around the wait for the teacher (host timers + device syncs; perturbs the overlap a little)"""
import os, sys, time, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
T = {}
orig_unl = trainer.forward_unlabel
def unl(data_u_list, features=None, job=None):
    T["enter_unlabel"] = time.perf_counter()
    if job is not None:
        job["thread"].join()
        T["teacher_joined"] = time.perf_counter()
    r = orig_unl(data_u_list, features, job)
    T["unlabel_launched"] = time.perf_counter()
    return r
trainer.forward_unlabel = unl
for i in range(3):
    il, tg, ul = batch(); trainer.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
for i in range(4):
    il, tg, ul = batch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.train_step(1410 + i, il, tg, ul)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step %.1f ms | host: to unlabel %.1f, wait for teacher %.1f, unlabel fwd launched +%.1f, rest of host +%.1f, GPU tail +%.1f" % (
        (t2 - t0) * 1e3, (T["enter_unlabel"] - t0) * 1e3, (T["teacher_joined"] - T["enter_unlabel"]) * 1e3,
        (T["unlabel_launched"] - T["teacher_joined"]) * 1e3, (t1 - T["unlabel_launched"]) * 1e3, (t2 - t1) * 1e3))
