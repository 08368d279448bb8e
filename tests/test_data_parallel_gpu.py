"""SURVEY.md 8(e) on the one GPU this box has: two ranks of the REAL detector's train_step on the same device, exchanging
through torch.distributed (gloo carries the device tensors; RCCL refuses two ranks on one GPU).  See tests/dp_worker.py
for what each rank checks; N > 1 on RCCL over xGMI is the driver's SCALE run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _two_rank_step(tmp_path, backend, port):
    env = dict(os.environ, MMT_BUCKETED_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", MMT_DP_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    return [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(2)]


@pytest.mark.skipif(_n_gpus() < 2, reason="RCCL needs one GPU per rank: this box has fewer than 2")
def test_two_rank_train_step_rccl(tmp_path):
    """the same checks as below on the REAL configs[3] path: two ranks, one GPU each, the nccl (= RCCL) backend -- exchanged
    gradient == mean of the per-rank gradients, same collective sequence on a rank that skips its consistency branch, identical
    students / teachers after a real step, the checksum collective (skipped only where there is one GPU)"""
    res = _two_rank_step(tmp_path, "nccl", 29733)
    for r in res:
        assert r["backend"] == "nccl" and r["device_collectives"], r
        assert r["grad_vs_mean"] < 1e-3 and r["skip_grad_vs_mean"] < 1e-3, r
        assert r["seq_equal_when_skipping"] and r["seq_equal_across_ranks"], r
        assert r["teacher_checksums_equal"] and r["student_checksums_equal"] and r["teacher_moved"], r
        assert r["check_passes"] and r["check_detects_divergence"], r
        assert r["skip_step_students_equal"] and r["skip_step_teachers_equal"] and r["skip_step_adaptor_moved"], r


def test_bench_gpus_n_never_runs_on_fewer_ranks():
    """`python bench.py --gpus N` outside a launcher starts N ranks itself -- or refuses: more GPUs requested than the node has
    exits non-zero before anything is timed, and a launcher that started a different number of ranks is refused too (a line
    labelled n_gpus = N is never produced by fewer ranks)"""
    n = _n_gpus()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"GPU(s) are visible" in p.stderr, p.stderr.decode()[-2000:]
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr, p.stderr.decode()[-2000:]


@pytest.mark.skipif(_n_gpus() < 2, reason="one process per GPU: this box has fewer than 2")
def test_bench_two_ranks_self_spawned():
    """`python bench.py --gpus 2` with no launcher: two RCCL ranks, the line says so and carries the exchange trace"""
    env = dict(os.environ, MMT_BENCH_NO_FP32_LEG="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--profile-steps", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"].startswith("nccl") and out["value"] > 0
    assert out["dist_trace"]["world"] == 2 and out["config"]["parallelism"] == "dp2"


def test_two_rank_train_step_on_one_device(tmp_path):
    res = _two_rank_step(tmp_path, "gloo", 29731)
    for r in res:
        # the two gradients come from different runs of the same step: ROIAlign / split-K atomics order only
        assert r["grad_vs_mean"] < 1e-3, r
        assert r["skip_grad_vs_mean"] < 1e-3, r
        assert r["pieces_early"] >= 3, r                      # FPN + heads, layer4, layer3 (, layer2) went out from hooks
        assert r["seq_equal_when_skipping"] and r["seq_equal_across_ranks"], r
        assert r["teacher_checksums_equal"] and r["student_checksums_equal"] and r["teacher_moved"], r
        assert r["check_passes"] and r["check_detects_divergence"], r
        # one rank skipped its consistency branch in a real step: the update set is the union over ranks, nobody drifts
        assert r["skip_step_students_equal"] and r["skip_step_teachers_equal"] and r["skip_step_adaptor_moved"], r
        assert r["reduced_keys"] == r["skip_keys_dp"], r
    # rank 1 skipped the consistency branch; its loss dict still carries the mt_* keys (zeros) for the logging reduce
    assert "mt_classifier" in res[1]["skip_keys"] and "mt_fg_loss" in res[1]["skip_keys"], res[1]
    assert res[0]["skip_keys"] == res[1]["skip_keys"]


def test_rccl_world1_bench_with_bucketed_exchange():
    """VERDICT r2 (next 7): multi-GPU readiness without the hardware.  `bench.py --gpus 1` under torch.distributed.run on the
    NCCL (= RCCL) backend at world size 1 (MMT_FORCE_DIST=1) with the bucketed, overlapped gradient exchange installed:
    the JSON line comes out, the collective sequence covers the flat gradient exactly once in the fixed order, the stage
    pieces go out BEFORE the backward pass has ended (the overlap survives RCCL's own streams: teacher stream priority -1),
    and the per-piece issue / arrival times are reported (what the first real 8-GPU run will be read by)."""
    env = dict(os.environ, MMT_FORCE_DIST="1", MMT_DIST_TRACE="1", MMT_BUCKETED_ALLREDUCE="1", MMT_BENCH_NO_FP32_LEG="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--profile-steps", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["consistency_branch_skipped_steps"] == 0
    assert out["rccl_ranks"] == 1 and out["backend"].startswith("nccl")
    tr = out["dist_trace"]
    assert tr["backend"] == "nccl" and tr["teacher_stream_priority"] == -1
    spans = sorted(tuple(q["range"]) for q in tr["pieces"])
    pos = 0
    for lo, hi in spans:                       # exact cover of the flat gradient, nothing twice
        assert lo == pos, spans
        pos = hi
    assert abs(pos * 4 / 1e6 - tr["grad_mbytes"]) < 0.1
    assert tr["pieces_sent_before_backward_end"] >= 3, tr     # FPN + heads, layer4, layer3 (, layer2) from the stage hooks
    early = tr["pieces"][:tr["pieces_sent_before_backward_end"]]
    assert all(q["issued_ms"] < tr["backward_end_ms"] for q in early), tr
    # most of the gradient is on the wire before backward ends: what is left for after it is layer2/layer1 + the biases
    assert sum(q["mbytes"] for q in early) > 0.7 * tr["grad_mbytes"], tr
    assert all(q["arrived_ms"] >= q["issued_ms"] for q in tr["pieces"])
