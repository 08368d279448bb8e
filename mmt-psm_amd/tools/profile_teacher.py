"""Tuning aid: GPU time of the sub-phases of forward_teacher and of the supervised heads."""
import os
import sys
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.cuda.set_device(0)
cfg, tr, batch = bench.build(torch.device("cuda", 0), 0)
T = tr.teacher
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        marks.append((label, e0, e1))
        return r
    setattr(obj, name, w)


wrap(T, "extract_aug_feat", "T backbone (8 views)")
wrap(T.rpn, "forward", "T coarse: rpn (head+postproc)")
wrap(T.rpn.head, "forward", "  rpn head convs")
wrap(T.rpn.box_selector_test, "forward", "  rpn postproc test")
wrap(T.rpn.box_selector_train, "forward", "  rpn postproc train-cfg")
wrap(T.box_heads.box, "forward", "T coarse: box head + postproc")
wrap(T.box_heads.box.post_processor, "forward", "  box postproc")
wrap(T.mask_heads.mask, "forward", "T coarse: mask head + paste")
wrap(T.rpn, "forward_teacher", "T rpn.forward_teacher")
wrap(T, "get_emb_feature", "T adaptors")
wrap(T.box_heads.box, "forward_teacher", "T box head x4 views")
wrap(T.box_heads.box.loss_evaluator, "subsample", "  subsample")
S = tr.student
wrap(S.rpn, "forward", "S rpn (head+postproc+loss)")
wrap(S.rpn.box_selector_train, "forward", "  S rpn postproc")
wrap(S.rpn.loss_evaluator, "__call__", "  S rpn loss")
wrap(S.box_heads.box, "forward", "S box head + loss")
wrap(S.mask_heads.mask, "forward", "S mask head + loss")
S.rpn.loss_evaluator = S.rpn.loss_evaluator
for i in range(3):
    marks.clear()
    il, tg, ul = batch()
    tr.train_step(1400 + i, il, tg, ul)
torch.cuda.synchronize()
for label, a, b in marks:
    print("%-36s %8.2f ms" % (label, a.elapsed_time(b)))
