"""Worker of tests/test_data_parallel_gpu.py: one rank of a 2-process data-parallel job on ONE GPU (the box has one), with
the gloo backend carrying the device tensors -- RCCL refuses two ranks on one device.  Runs the real detector's
`train_step` (engine/MTtrainer.py) with the bucketed, overlapped gradient exchange and checks SURVEY 8(e)'s pins:

  * the exchanged gradient == the mean of the two ranks' single-rank gradients (computed before the process group exists);
  * after a real step (SGD + EMA) both ranks hold the same student and the same teacher, and the checksum all-reduce
    (check_teacher_identity) passes; a corrupted teacher on one rank makes it raise on BOTH ranks;
  * a rank whose teacher finds no boxes (consistency branch skipped there) issues the same collective sequence: no hang,
    and the result is still the mean of what each rank computed.

Usage: python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_worker.py OUTDIR"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # MMT_DP_BACKEND=nccl (a node with >= 2 GPUs): one GPU per rank, RCCL over xGMI -- the real configs[3] path; default: both
    # ranks on device 0 with gloo carrying the device tensors (RCCL refuses two ranks on one GPU)
    backend = os.environ.get("MMT_DP_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import bench
    from maskrcnn_benchmark.engine import MTtrainer as MT
    cfg, trainer, batch = bench.build(dev, rank, crop=160, n_inst=4)   # per-rank data: seeds 1234 + rank / 4321 + rank
    res = {"rank": rank}

    def snapshot():
        return (trainer.flat_s.data.clone(), trainer.flat_t.data.clone(), trainer.flat_s.momentum.clone(),
                trainer.scheduler.last_epoch, trainer.optimizer.lr_factor)

    def restore(sn):
        trainer.flat_s.data.copy_(sn[0]); trainer.flat_t.data.copy_(sn[1]); trainer.flat_s.momentum.copy_(sn[2])
        trainer.scheduler.last_epoch, trainer.optimizer.lr_factor = sn[3], sn[4]
        trainer.flat_s.refresh_planes(); trainer.flat_t.refresh_planes()

    def step(update, seed=5):
        trainer.seed_rng(seed + 100 * rank)
        il, tg, ul = batch()
        keep = (trainer.optimizer.step, trainer.update_teacher)
        if not update:
            trainer.optimizer.step, trainer.update_teacher = (lambda: None), (lambda it: None)
        try:
            losses = trainer.train_step(1400, il, tg, ul)
        finally:
            trainer.optimizer.step, trainer.update_teacher = keep
        torch.cuda.synchronize()
        return {k: float(v) for k, v in losses.items()}, trainer.flat_s.grad.clone()

    sn = snapshot()
    step(False)                       # warm-up
    restore(sn)
    _, g_local = step(False)          # this rank's own gradient, no process group yet
    restore(sn)
    pp = trainer.teacher.box_heads.box.post_processor
    thr = pp.score_thresh
    if rank == 1:
        pp.score_thresh = 2.0         # the teacher finds nothing on this rank -> the consistency branch is skipped
    l_skip, g_local_skip = step(False)
    pp.score_thresh = thr
    restore(sn)
    res["skip_keys"] = sorted(l_skip)

    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res["backend"] = dist.get_backend()
    device_collectives = True
    try:
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t[0]) == world
    except Exception as e:            # a gloo build without device support: stage through the host, same call sequence
        device_collectives = False
        real = dist.all_reduce

        class _Done(object):
            def wait(self):
                return True

        def staged(t, op=dist.ReduceOp.SUM, async_op=False, group=None):
            if t.is_cuda:
                h = t.detach().cpu()
                real(h, op=op)
                t.copy_(h)
                return _Done() if async_op else None
            return real(t, op=op, async_op=async_op)

        real_reduce = dist.reduce

        def staged_reduce(t, dst, op=dist.ReduceOp.SUM, group=None):
            if t.is_cuda:
                h = t.detach().cpu()
                real_reduce(h, dst, op=op)
                t.copy_(h)
                return None
            return real_reduce(t, dst, op=op)

        dist.all_reduce, dist.reduce = staged, staged_reduce
        res["gloo_device_error"] = repr(e)[:200]
    res["device_collectives"] = device_collectives

    def gathered(t):
        if backend == "nccl":   # RCCL moves device tensors only
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            return [p.cpu() for p in parts]
        parts = [torch.zeros_like(t.cpu()) for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return parts

    # ---- (1) exchanged gradient == mean of the single-rank gradients
    assert trainer._bucketed_allreduce() is not None, "the bucketed exchange must be the one that runs"
    calls = []
    orig_send = MT.BucketedAllReduce._send

    def counting_send(self, lo, hi):
        calls.append((lo, hi))
        return orig_send(self, lo, hi)

    MT.BucketedAllReduce._send = counting_send
    _, g_dp = step(False)
    restore(sn)
    seq_normal = list(calls)
    parts = gathered(g_local)
    mean = sum(parts) / world
    err = (g_dp.cpu() - mean).abs().max().item() / mean.abs().max().item()
    res["grad_vs_mean"] = err
    res["collectives"] = len(seq_normal)
    res["pieces_early"] = len(seq_normal) - len(trainer._bucketed.rest)

    # ---- (2) one rank skips its consistency branch: same sequence of collectives everywhere, result still the mean
    del calls[:]
    if rank == 1:
        pp.score_thresh = 2.0
    l2, g_dp_skip = step(False)
    pp.score_thresh = thr
    restore(sn)
    res["seq_equal_when_skipping"] = calls == seq_normal
    seqs = [None] * world
    dist.all_gather_object(seqs, list(calls))
    res["seq_equal_across_ranks"] = all(s == seqs[0] for s in seqs)
    parts = gathered(g_local_skip)
    mean = sum(parts) / world
    res["skip_grad_vs_mean"] = (g_dp_skip.cpu() - mean).abs().max().item() / mean.abs().max().item()
    res["skip_keys_dp"] = sorted(l2)
    red = MT.reduce_loss_dict({k: torch.tensor(v, device=dev) for k, v in l2.items()})   # stacks the same keys everywhere
    res["reduced_keys"] = sorted(red)
    MT.BucketedAllReduce._send = orig_send

    # ---- (2b) ADVICE r2: a REAL step (SGD + EMA) with rank 1 skipping its consistency branch.  Rank 1 never touched the hint
    # adaptors, rank 0 did; the update set is the union over ranks (MTtrainer.sync_touched), so both students move the same
    if rank == 1:
        pp.score_thresh = 2.0
    step(True)
    pp.score_thresh = thr
    cs = gathered(torch.stack([MT.teacher_checksum(trainer.flat_t), MT.teacher_checksum(trainer.flat_s)]))
    res["skip_step_students_equal"] = bool(all(int(c[1]) == int(cs[0][1]) for c in cs))
    res["skip_step_teachers_equal"] = bool(all(int(c[0]) == int(cs[0][0]) for c in cs))
    o_, k_ = trainer.flat_s.index["hint_adaptor.adapter_1.weight"]
    res["skip_step_adaptor_moved"] = bool((trainer.flat_s.data[o_:o_ + k_] != sn[0][o_:o_ + k_]).any())
    restore(sn)

    # ---- (3) a real step: identical students and teachers afterwards, checksum collective passes
    step(True)
    cs = gathered(torch.stack([MT.teacher_checksum(trainer.flat_t), MT.teacher_checksum(trainer.flat_s)]))
    res["teacher_checksums_equal"] = bool(all(int(c[0]) == int(cs[0][0]) for c in cs))
    res["student_checksums_equal"] = bool(all(int(c[1]) == int(cs[0][1]) for c in cs))
    res["teacher_moved"] = bool((trainer.flat_t.data != sn[1]).any())
    res["check_passes"] = bool(MT.check_teacher_identity(trainer.flat_t))
    if rank == 1:
        trainer.flat_t.data[12345] += 1.0e-3      # a silent divergence on one rank
    try:
        MT.check_teacher_identity(trainer.flat_t)
        res["check_detects_divergence"] = False
    except RuntimeError:
        res["check_detects_divergence"] = True
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
