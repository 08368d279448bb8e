import os, sys, torch, logging
sys.path.insert(0, "/root/repo")
logging.basicConfig(level=logging.INFO)
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
for i in range(30):
    il, tg, ul = batch()
    l = trainer.train_step(1400 + i, il, tg, ul)
    print(i, " ".join("%s=%.4g" % (k, float(v)) for k, v in sorted(l.items())))
