"""Parity at BASELINE.json's FULL sizes (1000x1000 crops -> 1024x1024, 256-channel pyramid, 2000 proposals per
level) through size-independent properties, where running the CPU oracle would take minutes:
linearity, idempotence, flip-equivariance, batch-invariance, fixed points, and flat-vs-autograd gradient agreement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from maskrcnn_benchmark import _hip
    _hip.lib()
    return _hip


def cl(x):
    return x.contiguous(memory_format=torch.channels_last)


def test_nms_fullsize_idempotent_and_separated(hip):
    g = torch.Generator().manual_seed(1)
    segs, off = [], [0]
    for n in (2000, 2000, 2000, 2000, 768, 2000, 2000, 2000, 2000, 768):  # 2 images x 5 levels at 1024^2
        xy = torch.rand(n, 2, generator=g) * 900
        wh = torch.rand(n, 2, generator=g) * 160 + 8
        segs.append(torch.cat([xy, xy + wh], 1))
        off.append(off[-1] + n)
    boxes = torch.cat(segs).cuda()
    seg = torch.tensor(off, dtype=torch.int32).cuda()
    keep, cnt = hip.nms_batched(boxes, seg, 2000, 0.7)
    from maskrcnn_benchmark.structures.boxlist_ops import box_iou_tensor
    off2, kept_boxes = [0], []
    for i in range(10):
        k = keep[i, :int(cnt[i])].long()
        assert (k[1:] > k[:-1]).all()  # ascending positions == descending score
        b = boxes[off[i]:off[i + 1]][k]
        a = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        iou = box_iou_tensor(b, a, b, a)
        iou.fill_diagonal_(0)
        assert iou.max().item() < 0.7  # no surviving pair overlaps >= thr
        kept_boxes.append(b)
        off2.append(off2[-1] + len(k))
    keep2, cnt2 = hip.nms_batched(torch.cat(kept_boxes), torch.tensor(off2, dtype=torch.int32).cuda(), 2000, 0.7)
    for i in range(10):  # NMS of an NMS result keeps everything
        assert int(cnt2[i]) == off2[i + 1] - off2[i]


def test_roi_align_fullsize_linearity_and_flip(hip):
    g = torch.Generator().manual_seed(2)
    shapes = [(2, 256, 256 >> l, 256 >> l) for l in range(4)]
    fx = [cl(torch.randn(s, generator=g).cuda()) for s in shapes]
    fy = [cl(torch.randn(s, generator=g).cuda()) for s in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    K = 1024
    xy = torch.rand(K, 2, generator=g) * 800
    wh = torch.rand(K, 2, generator=g) * 220 + 4
    rois = torch.cat([(torch.arange(K) % 2).float()[:, None], xy, xy + wh], 1).cuda()
    s = torch.sqrt((wh[:, 0] + 1) * (wh[:, 1] + 1))
    lv = torch.clamp(torch.floor(4 + torch.log2(s / 224 + 1e-6)), 2, 5).int().cuda() - 2
    a = hip.roi_align_forward(fx, scales, rois, lv, 7, 7, 2)
    b = hip.roi_align_forward(fy, scales, rois, lv, 7, 7, 2)
    c = hip.roi_align_forward([2.0 * x - 0.5 * y for x, y in zip(fx, fy)], scales, rois, lv, 7, 7, 2)
    assert (c - (2.0 * a - 0.5 * b)).abs().max().item() < 2e-5
    # mirror symmetry: features mirrored along x, ROI mapped by the continuous coordinate x' = W_l-1 - x*scale
    # (i.e. image x' = Wimg - 1/scale - x, level dependent) give the mirrored pooled grid; borders excluded because
    # the reference clamps asymmetrically there
    Wimg = 1024
    inv = torch.tensor([1.0 / sc for sc in scales]).cuda()[lv.long()]
    rc = rois.clone()
    rc[:, 1] = Wimg - inv - rois[:, 3]
    rc[:, 3] = Wimg - inv - rois[:, 1]
    m2 = hip.roi_align_forward([torch.flip(x, (3,)) for x in fx], scales, rc, lv, 7, 7, 2)
    inner = (rois[:, 1] > 80) & (rois[:, 3] < Wimg - 80)
    assert inner.sum() > 100
    err = (torch.flip(m2, (3,)) - a)[inner].abs().max().item()
    assert err < 1e-3, err


def test_conv_fullsize_linearity_and_batch_invariance(hip):
    g = torch.Generator().manual_seed(3)
    w = cl((torch.randn(256, 256, 3, 3, generator=g) * 0.02).cuda())
    x = cl(torch.randn(2, 256, 256, 256, generator=g).cuda())
    y = cl(torch.randn(2, 256, 256, 256, generator=g).cuda())
    a, b = hip.conv_forward(x, w, pad=1), hip.conv_forward(y, w, pad=1)
    c = hip.conv_forward(3.0 * x - y, w, pad=1)
    assert (c - (3.0 * a - b)).abs().max().item() < 5e-4 * a.abs().max().item()
    # the same image gives bit-identical outputs whatever batch it sits in (what lets the engine batch the teacher
    # views and the student crops through one backbone pass)
    big = hip.conv_forward(cl(torch.cat([y, x, y, x], 0)), w, pad=1)
    assert torch.equal(big[2:4], a) and torch.equal(big[6:8], a) and torch.equal(big[0:2], b)


def test_ema_sgd_fixed_points_fullsize(hip):
    n = 44092257
    s = torch.randn(n, device="cuda")
    t = s.clone()
    hip.ema_update(t, s, 0.99)
    assert (t - s).abs().max().item() < 1e-6  # teacher == student is a fixed point
    p, buf = s.clone(), torch.zeros(n, device="cuda")
    hip.sgd_momentum(p, torch.zeros(n, device="cuda"), buf, 0.01, 0.0, 0.9, True)
    assert torch.equal(p, s)  # zero gradient, no decay: no movement


def test_flat_gradients_equal_autograd_gradients(synth, weights):
    """direct split-K accumulation into the flat gradient buffer == gradients returned through autograd"""
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.engine.flat import flatten_model
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    from maskrcnn_benchmark.structures.image_list import to_image_list
    cfg = make_default_cfg()
    imgs, tgs = synth.make_labeled(2, 192, 5, seed=7)

    def targets():
        out = []
        for t in tgs:
            b = BoxList(t["boxes"].cuda(), t["size"], "xyxy")
            b.add_field("labels", t["labels"].cuda())
            b.add_field("masks", SegmentationMask([[p for p in inst] for inst in t["polys"]], t["size"], mode="poly"))
            out.append(b)
        return out

    grads = []
    for flat in (False, True):
        m = build_detection_model(cfg, is_student=True)
        m.load_state_dict(weights, strict=False)
        m.cuda().train()
        if flat:
            flatten_model(m)
        torch.manual_seed(5)
        loss = sum(m(to_image_list(list(imgs.cuda()), 32), targets()).values())
        loss.backward()
        grads.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert set(grads[0]) <= set(grads[1])  # the flat model pre-allocates (zero) gradients for unused parameters too
    for k in grads[1]:
        if k not in grads[0]:
            assert grads[1][k].abs().max().item() == 0.0, k
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert (a - b).abs().max().item() <= 2e-4 * (a.abs().max().item() + 1e-9), k


def test_teacher_batched_views_equal_unbatched(synth, weights):
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.image_list import to_image_list
    cfg = make_default_cfg()
    t = build_detection_model(cfg, is_teacher=True)
    t.load_state_dict(weights, strict=False)
    t.cuda().eval()
    unl = synth.make_unlabeled(2, 160, 3, seed=11)
    with torch.no_grad():
        ils = [to_image_list(list(u.cuda()), 32) for u in unl[:2]]
        feats = t.extract_aug_feat(ils)
        ref = []
        for u in unl[:2]:
            x = to_image_list(list(u.cuda()), 32).tensors
            ref += [t.backbone(x), t.backbone(torch.flip(x, (3,)))]
    for a, b in zip(feats, ref):
        for la, lb in zip(a, b):
            # same per-image arithmetic; the deep, few-tile layers pick their number of K ranges (split-K) from the tile count,
            # so the summation ORDER may differ between the batched and the single-view pass: equal to fp32 rounding
            assert (la - lb).abs().max().item() <= 2e-5 * lb.abs().max().item()


def test_bucketed_allreduce_covers_every_gradient_exactly_once(monkeypatch):
    """engine/MTtrainer.py::BucketedAllReduce starts the gradient exchange piece by piece while backward is running.
    With torch.distributed replaced by a fake 2-rank world whose "all-reduce" doubles its buffer in place, the final
    gradient must equal the plain one: a piece sent too early (before its last accumulation) or a piece sent twice or
    never would show up as a mismatch."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
    import bench
    from maskrcnn_benchmark.engine import MTtrainer as MT
    cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0)
    # the overlapped default schedule: each model draws from its own generator, so two runs of a step draw the same samples

    def grads(step):
        il, tg, ul = batch()
        trainer.seed_rng(7)                            # the student's and the teacher's own random streams
        opt_step = trainer.optimizer.step
        trainer.optimizer.step = lambda: None          # keep the weights: same forward both times
        upd = trainer.update_teacher
        trainer.update_teacher = lambda it: None
        try:
            trainer.train_step(step, il, tg, ul)
        finally:
            trainer.optimizer.step, trainer.update_teacher = opt_step, upd
        torch.cuda.synchronize()
        return trainer.flat_s.grad.clone()

    grads(1400)                       # warm-up (allocator, caches)
    ref = grads(1401)
    calls = []

    class Work(object):
        """what a process-group work handle does: `wait` orders the CALLER's stream behind the collective, which ran on the stream
        it was issued from (the weight-gradient side stream, engine/MTtrainer.py::BucketedAllReduce._send)"""

        def __init__(self):
            self.ev = torch.cuda.Event()
            self.ev.record()

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    def fake_all_reduce(t, op=None, async_op=False, group=None):
        if t.dtype != torch.float32:   # MTtrainer.sync_touched's per-parameter flags (MAX): equal on the two fake ranks
            return Work() if async_op else None
        calls.append((t.data_ptr(), t.numel()))
        t.mul_(2.0)
        return Work() if async_op else None   # (the event is recorded behind the doubling, on the issuing stream)

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(dist, "new_group", lambda *a, **k: None)   # (sync_touched exchanges its flags on a communicator of its own)
    monkeypatch.setattr(MT, "get_world_size", lambda: 2)
    MT._FLAG_SYNC.clear()
    got = grads(1401)
    base = trainer.flat_s.grad.data_ptr()
    spans = sorted(((p - base) // 4, (p - base) // 4 + n) for p, n in calls)
    assert len(spans) >= 5, spans     # heads, FPN, layer4, layer3 went out early; the rest at the end
    pos = 0
    for lo, hi in spans:              # exact cover, no overlap
        assert lo == pos, spans
        pos = hi
    assert pos == trainer.flat_s.grad.numel()
    # same seeds, same draws: the two gradients differ only by the order of the fp32 atomics in ROIAlign backward
    assert (got - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    trainer._bucketed = None
    MT._FLAG_SYNC.clear()
