"""BASELINE configs[4] as ONE tested workload on one GPU: IR-Net (relation NMS + mask relation) x bf16
products with bf16 activation storage in the backbone + FPN (`bench.py --bf16 --irnet`) x one WHOLE mean-teacher iteration
[A]-[E] of engine/MTtrainer.py, against the fp32 CPU oracle's trainer (oracle/model.py::Trainer with oracle/irnet.py; pinned
to the reference by tests/golden/model160_irnet.npz) at bf16 tolerances:

  * every loss of the step (supervised, nms_loss, two-stage loss_seg, mt_fg_loss, mt_classifier) within 4e-3 relative (measured
    <= 1.6e-3; the bar is stated next to BF16_TOL);
  * the SGD update of a subset of tensors along the whole path (FPN, RPN head, fc7, mask head, relation modules, hint adaptor,
    a layer3 weight) against the oracle's at 0.08 relative in the L2 norm (measured <= 0.056; relation-NMS parameters 0.75: their gradient hangs on
    discrete selections) -- the gradient is d(update): weight decay is 1e-4;
  * the teacher after the EMA;
  * only the RANDOM draws are replayed; a proposal / detection list stays the product's own when ALL its discrete decisions agree
    with the oracle's (same count, boxes within 1 px) and is substituted -- and listed -- when bf16 arithmetic flipped one (with
    thousands of candidates per list that is the rule); as SETS the lists must still be > 90 % the oracle's.
And the bench command of that configuration runs end to end (a few steps at the full 1000 x 1000 size)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import GOLD, ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)
# The bar of this configuration.  There is no bf16 oracle to be bit-compared with: the reference has no bf16 path
# (tools/train_mean_teacher.py runs fp32), so configs[4]'s "bf16 MFMA path" is judged against the fp32 oracle with a stated error
# budget -- a bf16 product carries 2^-9 relative error per operand, a loss is a mean over >= 10^4 such terms, and the discrete
# decisions bf16 flips are replayed: every loss within 4e-3 relative = 2.5 x the largest deviation measured since round 3 (1.6e-3).
# What stays on the fp32-grade arithmetic in this configuration and why: the heads (RPN predictors, fc6 / fc7, mask head on pooled
# features), every weight gradient and the losses -- their inputs are fp32 tensors produced by fp32 ROIAlign / selection kernels, their
# time is 15 % of the step, and the relation modules' gradients hang on discrete selections that bf16 already perturbs (below).
BF16_TOL = 4e-3   # (round 5: 1e-2; round 3: 5e-2)


@pytest.fixture()
def bf16_mode():
    from maskrcnn_benchmark import _hip as H
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(1)
    H.set_bf16_storage(True)
    yield H
    H.set_bf16_storage(False)
    H.set_conv_precision(prev)


def test_irnet_bf16_storage_full_mean_teacher_step_vs_fp32_oracle(bf16_mode, synth):
    import bench
    from oracle import model as om
    from maskrcnn_benchmark.utils.replay import Replay
    from test_train_step_gpu import _load, _oracle_targets, _param
    ss = json.load(open(os.path.join(GOLD, "state_shapes_irnet.json")))
    weights = synth.make_weights(ss["shapes"], seed=0)
    cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, irnet=True, crop=160, n_inst=4)
    assert cfg.MODEL.RELATION_NMS.USE_RELATION_NMS and cfg.MODEL.RELATION_MASK.USE_RELATION
    _load(trainer, weights)
    ot = om.Trainer(weights, om.default_cfg(relation=True, nms_loss_w=cfg.MODEL.RELATION_NMS.LOSS), ss["trainable"], ss["param_order"])
    ot.last_epoch = trainer.scheduler.last_epoch
    imgs, tgs = synth.make_labeled(2, 160, 4, seed=1234)
    unl = synth.make_unlabeled(2, 160, 3, seed=4321)
    ref_losses, (ta, tb, tc) = ot.step(1400, imgs, _oracle_targets(om, tgs), unl, seeds=(99, 100, 101))
    names = ["backbone.fpn.fpn_layer2.weight", "backbone.fpn.fpn_inner4.weight", "backbone.body.layer3.2.conv2.weight",
             "rpn.head.conv.weight", "box_heads.box.feature_extractor.fc7.weight", "mask_heads.mask.feature_extractor.mask_fcn4.weight",
             "relation_nms.relation_module.WQ.weight", "relation_nms.nms_rank_fc.weight",
             "mask_heads.mask.mask_relation_module.appearance_feature_extractor.mask_fcn2.weight", "hint_adaptor.adapter_2.weight"]
    names = [n for n in names if n in ss["trainable"]]
    assert len(names) >= 8, names
    before_s = {n: _param(trainer.flat_s, trainer.student, n) for n in names}
    before_t = {n: _param(trainer.flat_t, trainer.teacher, n) for n in names}
    stu = dict(ta)
    stu["dropout"] = list(ta["dropout"]) + list(tc.get("dropout", []))
    for k, v in tc.items():
        stu.setdefault(k, v)
    trainer.student.set_replay(Replay(stu, substitute_lists="where_different"))
    trainer.teacher.set_replay(Replay(tb, substitute_lists="where_different"))
    trainer.student.taps, trainer.teacher.taps = {}, {}
    try:
        il, tg, ul = batch()
        with torch.no_grad():   # the pyramid really is stored as bf16 in this configuration
            seen = [t.dtype for t in trainer.student.backbone(il.tensors.cuda())]
        losses = trainer.train_step(1400, il, tg, ul)
        torch.cuda.synchronize()
    finally:
        trainer.student.set_replay(None)
        trainer.teacher.set_replay(None)
        own = dict(trainer.student.taps)
        own.update({"T." + k: v for k, v in trainer.teacher.taps.items()})
        trainer.student.taps = trainer.teacher.taps = None
    assert seen and all(d == torch.bfloat16 for d in seen), seen
    subst = sorted(k[:-len("_substituted")] for k, v in own.items() if k.endswith("_substituted") and v)
    kept = sorted(k[:-len("_substituted")] for k, v in own.items() if k.endswith("_substituted") and not v)
    agree = {k[:-len("_agreement")]: round(v, 4) for k, v in own.items() if k.endswith("_agreement")}
    print("lists kept as computed:", kept, "| substituted (a bf16 decision differed somewhere in the list):", subst,
          "| share of the oracle's boxes the product produced too:", agree)
    # thousands of candidates per list: with bf16 products SOME near-threshold decision flips in every list, so the lists are
    # substituted (the recorded sampler positions must refer to the same boxes) -- but as sets they are nearly the oracle's
    # (the detections are the few boxes that survive IR-Net's learned duplicate removal at a score threshold: a handful, each a
    # threshold decision on a regressed value -- 0.6; the proposal lists 0.9)
    assert agree and all(v > (0.6 if "detections" in k else 0.9) for k, v in agree.items()), agree
    assert set(losses) == set(ref_losses), (sorted(losses), sorted(ref_losses))
    dev = {k: abs(float(losses[k]) - float(v)) / max(abs(float(v)), 1e-6) for k, v in ref_losses.items()}
    print("relative deviation of the losses:", {k: round(v, 5) for k, v in dev.items()})
    assert all(v == v and v < BF16_TOL for v in dev.values()), dev
    worst = {}
    for n in names:
        d_own = _param(trainer.flat_s, trainer.student, n) - before_s[n]
        d_ref = ot.s[n].detach() - weights[n]
        worst[n] = ((d_own - d_ref).norm() / d_ref.norm().clamp_min(1e-30)).item()
        assert d_ref.norm().item() > 0, n
        t_own = _param(trainer.flat_t, trainer.teacher, n) - before_t[n]
        t_ref = ot.t[n] - weights[n]
        assert (t_own - t_ref).norm().item() <= 0.2 * t_ref.norm().item() + 1e-6 * weights[n].norm().item(), n
    print("relative L2 error of the SGD update:", {k: round(v, 4) for k, v in worst.items()})
    # measured 0.01-0.06 on the trunk, the heads, the mask relation module and the adaptors.  The relation-NMS parameters hang on
    # DISCRETE selections (rank embedding of score-sorted boxes, top-40 attention partners, argmax label preparation,
    # relation_module.py:137-391): a bf16-flipped selection changes which rows receive a gradient at all (0.2-0.6 measured)
    assert all(v < (0.75 if n.startswith("relation_nms.") else 0.08) for n, v in worst.items()), worst


def test_bench_bf16_irnet_runs():
    """`python bench.py --bf16 --irnet` (BASELINE configs[4] on one GPU) end to end at the full size, a few steps"""
    env = dict(os.environ, MMT_BENCH_NO_FP32_LEG="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bf16", "--irnet", "--steps", "3", "--warmup", "2",
                        "--profile-steps", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["dtype"] == "bf16" and out["value"] > 0 and "IR-Net ON" in out["config"]["workload"]
    assert out["config"]["consistency_branch_skipped_steps"] == 0
    assert all(v == v for v in out["config"]["losses"].values()) and "nms_loss" in out["config"]["losses"]
