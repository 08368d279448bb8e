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


def test_two_rank_train_step_on_one_device(tmp_path):
    env = dict(os.environ, MMT_BUCKETED_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    res = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(2)]
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
