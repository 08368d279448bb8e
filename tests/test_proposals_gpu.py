"""SURVEY.md rows a7 / a22 as TESTED stages: the product's own proposal and detection lists against the CPU oracle
(oracle/model.py::rpn_postprocess / box_postprocess, restating rpn/inference.py:78-243 and
box_head/inference.py:36-145), at sizes where every cap binds:

  * 2 images padded to 1024 x 1024 -> 5 levels of (256, 128, 64, 32, 16)^2 x 3 anchors: the pre-NMS top-2000 binds on
    three levels (top-1000 on three levels for the test selector), the NMS removes hundreds of boxes per segment, the
    batch-wide top-2000 (training), the per-image top-2000 / top-1000 (eval) and add_gt_proposals all take effect;
  * 1000 proposals per image through the detection post-processor with far more than 200 survivors of the per-class
    NMS, so the `kthvalue` cut to DETECTIONS_PER_IMG binds.

The head outputs are synthetic (so that the sizes above cost seconds on the CPU side) but go through the product's own
`RPNPostProcessor` / `PostProcessor` modules exactly as the model calls them.  Scores are drawn WITHOUT ties: objectness
logits are logit(p) for distinct p on a 2^-18 grid, so that a last-ulp difference between the device's and the host's
sigmoid cannot reorder two candidates -- the reference leaves the order of equal scores unspecified.
Bar: identical counts, boxes <= 1e-3 px, scores <= 1e-6, integer fields exact, in identical order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from proposal_inputs import N_IMG, SIZE, PAD, GRIDS, A, head_outputs as _head_outputs, gt_boxes, box_head_inputs


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def _gt_targets(om, seed):
    out_o, out_p = [], []
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    for b in gt_boxes(seed):
        out_o.append(om.Boxes(b, (SIZE, SIZE), {"labels": torch.ones(12, dtype=torch.int64)}))
        out_p.append(BoxList(b.cuda(), (SIZE, SIZE), "xyxy"))
    return out_o, out_p


def _compare(own, ref, fields=("objectness",), int_fields=()):
    assert len(own) == len(ref)
    for o, r in zip(own, ref):
        assert len(o) == len(r), (len(o), len(r))
        np.testing.assert_allclose(o.bbox.cpu().numpy(), r.bbox.numpy(), rtol=0, atol=1e-3)
        for f in fields:
            np.testing.assert_allclose(o.get_field(f).cpu().numpy(), r.fields[f].numpy(), rtol=0, atol=1e-6)
        for f in int_fields:
            np.testing.assert_array_equal(o.get_field(f).cpu().numpy(), r.fields[f].numpy())


@pytest.fixture(scope="module")
def rpn():
    from maskrcnn_benchmark import _hip
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.rpn.rpn import build_rpn
    _hip.lib()
    cfg = make_default_cfg()
    return cfg, build_rpn(cfg, is_teacher=True).cuda()


def _anchors(rpn_module, om, ocfg):
    from maskrcnn_benchmark.structures.image_list import ImageList
    il = ImageList(torch.zeros(N_IMG, 3, PAD, PAD, device="cuda"), [(SIZE, SIZE)] * N_IMG)
    feats = [torch.zeros(N_IMG, 1, s, s, device="cuda") for s in GRIDS]
    a_p = rpn_module.anchor_generator(il, feats)
    a_o = om.make_anchors(ocfg, [(SIZE, SIZE)] * N_IMG, [(s, s) for s in GRIDS])
    for lp, lo in zip(a_p[0], a_o[0]):  # a6 on the way: anchors and visibility bit-exact at full size
        assert torch.equal(lp.bbox.cpu(), lo.bbox)
        assert torch.equal(lp.get_field("visibility").cpu(), lo.fields["visibility"])
    return a_p, a_o


def test_rpn_selector_training_caps_bind(rpn):
    """box_selector_train of a training module: pre-NMS top-2000 per level, NMS 0.7, batch-wide top-2000, + GT boxes"""
    from oracle import model as om
    cfg, m = rpn
    ocfg = om.default_cfg()
    obj, reg = _head_outputs(3)
    a_p, a_o = _anchors(m, om, ocfg)
    t_o, t_p = _gt_targets(om, 5)
    ref = om.rpn_postprocess(ocfg, a_o, obj, reg, True, True, t_o)
    sel = m.box_selector_train
    sel.train()
    with torch.no_grad():
        own = sel(a_p, [_cl(o) for o in obj], [_cl(r) for r in reg], t_p)
    n_ref = [len(r) for r in ref]
    assert sum(n_ref) == 2000 + 24 and min(n_ref) > 200, n_ref   # the batch-wide cap binds, both images contribute
    _compare(own, ref)
    # the caps really were active on the way
    k_pre = [min(2000, A * s * s) for s in GRIDS]
    assert k_pre[:3] == [2000, 2000, 2000] and k_pre[3:] == [2000, 768]


def test_rpn_selector_eval_and_teacher_fields(rpn):
    """the two selectors the teacher runs on pyramid 0 (generalized_rcnn.py:126,146): the TEST-config selector (pre/post
    1000, per-image top-1000) and the TRAIN-config selector of an eval() module (pre 2000, per-image top-2000, no GT)
    with the teacher's extra fields -- sharing one decode + NMS through `shared`, as forward_teacher does"""
    from oracle import model as om
    cfg, m = rpn
    ocfg = om.default_cfg()
    obj, reg = _head_outputs(4)
    a_p, a_o = _anchors(m, om, ocfg)
    ref_test = om.rpn_postprocess(ocfg, a_o, obj, reg, False, False)
    ref_teach = om.rpn_postprocess(ocfg, a_o, obj, reg, True, False, None, is_teacher=True)
    m.eval()
    shared = {"pre": max(m.box_selector_train.pre_nms_top_n, m.box_selector_test.pre_nms_top_n)}
    dobj, dreg = [_cl(o) for o in obj], [_cl(r) for r in reg]
    with torch.no_grad():
        own_test = m.box_selector_test(a_p, dobj, dreg, shared=shared)
        own_teach = m.box_selector_train(a_p, dobj, dreg, None, shared=shared)
        alone_test = m.box_selector_test(a_p, dobj, dreg)          # and without the shared candidates
    assert [len(r) for r in ref_test] == [1000, 1000]
    assert [len(r) for r in ref_teach] == [2000, 2000]
    _compare(own_test, ref_test)
    _compare(alone_test, ref_test)
    _compare(own_teach, ref_teach, fields=("objectness", "box_reg"), int_fields=("rpn_topk", "rpn_ancher_level"))
    m.train()


def test_rpn_selector_min_size(rpn):
    """RPN.MIN_SIZE > 0 (not the shipped value): small boxes are removed BEFORE the NMS, so they neither suppress nor
    take post-NMS slots (rpn/inference.py:124-129)"""
    from oracle import model as om
    from maskrcnn_benchmark.modeling.rpn.rpn import RPNPostProcessor
    cfg, m = rpn
    ocfg = om.default_cfg(rpn_min_size=24, post_nms_train=300)
    obj, reg = _head_outputs(6)
    for r in reg:
        r[:, 2::4] -= 1.0   # many narrow boxes
    a_p, a_o = _anchors(m, om, ocfg)
    ref = om.rpn_postprocess(ocfg, a_o, obj, reg, True, False)
    sel = RPNPostProcessor(2000, 300, 0.7, 24, fpn_post_nms_top_n=2000).cuda().eval()
    with torch.no_grad():
        own = sel(a_p, [_cl(o) for o in obj], [_cl(r) for r in reg])
    assert all(200 < len(r) <= 1500 for r in ref), [len(r) for r in ref]
    _compare(own, ref)


def test_box_postprocessor_more_than_200_detections():
    """PostProcessor.forward (box_head/inference.py:36-145) on 1000 proposals per image: softmax, decode (10,10,5,5),
    clip, per-class score > 0.05, NMS 0.5, `kthvalue` cut to 200"""
    from oracle import model as om
    from maskrcnn_benchmark import _hip
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.roi_heads.box_head.box_head import make_roi_box_post_processor
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    _hip.lib()
    ocfg = om.default_cfg()
    boxes, objs, logits, deltas = box_head_inputs(8, 1000)
    props_o, props_p = [], []
    for b, o in zip(boxes, objs):
        props_o.append(om.Boxes(b, (SIZE, SIZE), {"objectness": o}))
        p = BoxList(b.cuda(), (SIZE, SIZE), "xyxy")
        p.add_field("objectness", o.cuda())
        props_p.append(p)
    ref = om.box_postprocess(ocfg, logits, deltas, props_o)
    pp = make_roi_box_post_processor(make_default_cfg()).cuda()
    with torch.no_grad():
        own = pp((logits.cuda(), deltas.cuda()), props_p)
    assert [len(r) for r in ref] == [200, 200]
    # before the cut there were many more (the cap binds)
    before = om.box_postprocess(om.default_cfg(dets_per_img=10 ** 6), logits, deltas, props_o)
    assert min(len(r) for r in before) > 400, [len(r) for r in before]
    _compare(own, ref, fields=("scores",), int_fields=("labels",))
    # and the uncut list as well (per-class NMS of ~600 boxes per class)
    pp2 = make_roi_box_post_processor(make_default_cfg()).cuda()
    pp2.detections_per_img = 10 ** 6
    with torch.no_grad():
        own2 = pp2((logits.cuda(), deltas.cuda()), props_p)
    _compare(own2, before, fields=("scores",), int_fields=("labels",))
