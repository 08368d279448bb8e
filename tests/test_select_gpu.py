"""the selection kernels of the proposal pipeline (csrc/select.hip: mmt_rpn_gather_decode,
mmt_rpn_post_select, mmt_sample_fg_bg) bit for bit against the tensor formulations they replaced
(tests/tensor_formulations.py), at the full 1024 x 1024 sizes.  Scores / keys are drawn without ties: the order between
equal values is unspecified in the reference and "lower index first" in the kernels."""
import numpy as np
import pytest
import torch

import tensor_formulations as tf

pytestmark = pytest.mark.gpu

N, A = 2, 3
GRIDS = [256, 128, 64, 32, 16]


@pytest.fixture(scope="module")
def hip():
    from maskrcnn_benchmark import _hip
    _hip.lib()
    return _hip


def _heads(seed):
    g = torch.Generator().manual_seed(seed)
    heads = []
    for s in GRIDS:
        n = s * s * A
        # distinct logits per (image, level): a random permutation of a fine grid
        lg = torch.stack([(torch.randperm(n, generator=g).float() / n) * 12.0 - 8.0 for _ in range(N)]).view(N, s, s, A)
        rg = torch.randn(N, s, s, 4 * A, generator=g) * 0.5
        h = torch.cat([lg, rg], 3).permute(0, 3, 1, 2)          # (N, 5A, H, W) with NHWC memory
        heads.append(h.cuda())
    return heads


def _anchors():
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import make_anchor_generator
    ag = make_anchor_generator(make_default_cfg()).cuda()
    return ag.grid_anchors([(s, s) for s in GRIDS])


def test_rpn_gather_decode_and_post_select(hip):
    import math
    heads, anchors = _heads(3), _anchors()
    clip = math.log(1000.0 / 16)
    lims = torch.tensor([[999.0, 999.0], [899.0, 949.0]]).cuda()
    obj = [h[:, :A] for h in heads]
    reg = [h[:, A:] for h in heads]
    rb, rs, ri, rr, offs = tf.rpn_candidates(obj, reg, anchors, 2000, clip, lims)
    topks = [o.permute(0, 2, 3, 1).reshape(N, -1).topk(min(2000, o[0].numel()), dim=1, sorted=True)[1] for o in obj]
    for a, b in zip(topks, [ri[:, offs[l]:offs[l + 1]] for l in range(5)]):
        assert torch.equal(a, b)
    b, s, i, r, o2 = hip.rpn_gather_decode(heads, anchors, topks, A, clip, lims)
    assert o2 == offs and torch.equal(i, ri) and torch.equal(r, rr) and torch.equal(s, rs)
    # decode: device exp() in both, but the tensor formulation multiplies where the kernel does the same operations fused
    assert torch.equal(b, rb)
    sumk = offs[-1]
    seg = [n * sumk + offs[l] for n in range(N) for l in range(5)] + [N * sumk]
    keep, cnt = hip.nms_batched(b.view(-1, 4), torch.tensor(seg, dtype=torch.int32).cuda(), 2000, 0.7)
    g = torch.Generator().manual_seed(5)
    gt = (torch.rand(19, 4, generator=g) * 500).cuda()
    gt[:, 2:] += gt[:, :2]
    gt_off = torch.tensor([0, 12, 19], dtype=torch.int32).cuda()
    for (own_pre, post_n, fpn, training) in (([2000] * 4 + [768], 2000, 2000, True), ([1000] * 4 + [768], 1000, 1000, False),
                                             ([2000] * 4 + [768], 2000, 2000, False), ([2000] * 4 + [768], 300, 1000, True)):
        ref = tf.rpn_post_select(s, keep, cnt, offs, own_pre, post_n, fpn, training)
        cap = fpn + 12
        ob, osc, oi, orr, ol, oc = hip.rpn_post_select(b, s, i, r, keep, cnt, offs, own_pre, post_n, fpn, training, cap,
                                                       gt if training else None, gt_off if training else None)
        oc = oc.tolist()
        for n in range(N):
            ng = (12, 7)[n] if training else 0
            assert oc[n] == ref[n].numel() + ng, (oc, [x.numel() for x in ref])
            k = ref[n].numel()
            assert torch.equal(ob[n, :k], b[n][ref[n]]) and torch.equal(osc[n, :k], s[n][ref[n]])
            assert torch.equal(oi[n, :k], i[n][ref[n]]) and torch.equal(orr[n, :k], r[n][ref[n]])
            lvl = torch.bucketize(ref[n], torch.tensor(offs[1:], device="cuda"), right=True)
            assert torch.equal(ol[n, :k].long(), lvl)
            if training:
                g0 = (0, 12)[n]
                assert torch.equal(ob[n, k:k + ng], gt[g0:g0 + ng]) and bool((osc[n, k:k + ng] == 1).all())
        if training:
            assert sum(x.numel() for x in ref) == min(fpn, sum(x.numel() for x in ref))


@pytest.fixture(params=[True, False], ids=["wide", "one-block"])
def sampler_form(request, hip):
    """both forms of the sampler on every case: the many-blocks-wide one (mmt_sample_fg_bg_wide, the RPN's long vectors) and
    the one-block-per-image kernel (mmt_sample_fg_bg) -- the same masks bit for bit"""
    old = (hip.SAMPLE_WIDE, hip.SAMPLE_WIDE_MIN)
    hip.SAMPLE_WIDE, hip.SAMPLE_WIDE_MIN = request.param, 1
    yield request.param
    hip.SAMPLE_WIDE, hip.SAMPLE_WIDE_MIN = old


@pytest.mark.parametrize("dtype", [torch.float32, torch.int64])
def test_sample_fg_bg_kernel(hip, dtype, sampler_form):
    g = torch.Generator().manual_seed(11)
    lens = [261888, 261888, 2012, 37, 1500]
    labs, keys = [], []
    for n in lens:
        u = torch.rand(n, generator=g)
        lab = torch.where(u < 0.002, 1 + (torch.arange(n) % 2), torch.where(u < 0.9, 0, -1))
        if n == 37:
            lab[:] = 0                                   # fewer negatives than the batch, no positives
        if n == 1500:
            lab[:400] = 1                                # more positives than the quota
        labs.append(lab.to(dtype))
        keys.append(torch.randperm(n, generator=g).float() / n)   # distinct keys
    labels, kk = torch.cat(labs).cuda(), torch.cat(keys).cuda()
    off = torch.tensor(np.cumsum([0] + lens), dtype=torch.int32).cuda()
    pm, nm, cnt = hip.sample_fg_bg(labels, kk, off, 256, 128)
    o = 0
    for i, n in enumerate(lens):
        rp, rn = tf.sample_fg_bg(labels[o:o + n], kk[o:o + n], 256, 128)
        assert torch.equal(pm[o:o + n], rp) and torch.equal(nm[o:o + n], rn), i
        assert cnt[i].tolist() == [int(rp.sum()), int(rn.sum())]
        o += n
    assert cnt[3].tolist() == [0, 37] and cnt[4].tolist() == [128, 128]


def test_sample_fg_bg_any_key_distribution(hip, sampler_form):
    """keys that are NOT uniform (most below the kernel's pre-filter, or none): the exact select over the whole vector
    takes over -- same result as the tensor formulation"""
    g = torch.Generator().manual_seed(12)
    n = 20000
    lab = (torch.rand(n, generator=g) < 0.01).long()
    for keys in ((torch.randperm(n, generator=g).float() / n) * 1e-3,          # everything below tau
                 0.9 + (torch.randperm(n, generator=g).float() / n) * 0.09):   # nothing below tau
        off = torch.tensor([0, n], dtype=torch.int32).cuda()
        pm, nm, cnt = hip.sample_fg_bg(lab.cuda(), keys.cuda(), off, 256, 128)
        rp, rn = tf.sample_fg_bg(lab.cuda(), keys.cuda(), 256, 128)
        assert torch.equal(pm, rp) and torch.equal(nm, rn)


def test_roi_format_levels_kernel(hip):
    """mmt_roi_format_levels == convert_to_roi_format + LevelMapper, bit for bit, on boxes that straddle every level boundary
    (sizes swept densely around 112 / 224 / 448 px), degenerate boxes, an empty image, fixed-capacity views"""
    H = hip
    g = torch.Generator().manual_seed(5)
    boxes = []
    for n in (3000, 0, 1777, 1, 4096):
        side = torch.cat([torch.rand(n // 2, generator=g) * 900 + 1,
                          torch.tensor([112.0, 224.0, 448.0])[torch.randint(0, 3, (n - n // 2,), generator=g)]
                          * (1 + (torch.rand(n - n // 2, generator=g) - 0.5) * 1e-3)])
        asp = torch.exp((torch.rand(n, generator=g) - 0.5) * 1.5)
        w, h = side * asp.sqrt(), side / asp.sqrt()
        xy = torch.rand(n, 2, generator=g) * 500
        b = torch.stack([xy[:, 0], xy[:, 1], xy[:, 0] + w - 1, xy[:, 1] + h - 1], 1)
        if n > 10:
            b[3] = torch.tensor([5.0, 5.0, 5.0, 5.0])      # 1 x 1 px
            b[4] = torch.tensor([7.0, 9.0, 6.0, 8.0])      # x2 = x1 - 1: area 0
        cap = torch.zeros((n + 5, 4))
        cap[:n] = b
        boxes.append(cap.cuda()[:n])                       # a prefix view of a fixed-capacity tensor, as the RPN hands them over
    rois, lv = H.roi_format_levels(boxes, 224, 4, 1e-6, 2, 5)
    ref_rois, ref_lv = tf.roi_format_levels(boxes, 2, 5)
    assert torch.equal(rois, ref_rois)
    assert torch.equal(lv, ref_lv)
    assert set(lv.unique().tolist()) == {0, 1, 2, 3}


def test_rpn_topk_kernel(hip):
    """mmt_rpn_topk == torch.topk(sorted=True) per (image, level) on tie-free logits at the full sizes (k = 2000 on four levels,
    all 768 anchors of the coarsest), for k = 1000 as well; equal logits come out lower index first, and a segment of
    thousands of equal logits AT the threshold (a blank image) takes the exact 64-bit path"""
    H = hip
    heads = _heads(17)
    for pre in (2000, 1000):
        ks = [min(pre, h.shape[2] * h.shape[3] * A) for h in heads]
        got = H.rpn_topk(heads, ks, A)
        for h, k, t in zip(heads, ks, got):
            ref = h[:, :A].permute(0, 2, 3, 1).reshape(N, -1).topk(k, dim=1, sorted=True)[1]
            assert torch.equal(t, ref), (pre, k)
    # ties: quantised logits (every value ~50 times) and one image that is constant
    g = torch.Generator().manual_seed(3)
    th = []
    for s in (128, 32):
        lg = torch.randint(0, 1000, (N, s, s, A), generator=g).float() * 0.01 - 5.0
        lg[1] = 0.25
        h = torch.cat([lg, torch.zeros(N, s, s, 4 * A)], 3).permute(0, 3, 1, 2).cuda()
        th.append(h)
    ks = [2000, 2000]
    got = H.rpn_topk(th, ks, A)
    for h, k, t in zip(th, ks, got):
        flat = h[:, :A].permute(0, 2, 3, 1).reshape(N, -1)
        n = flat.shape[1]
        # reference with the kernel's tie rule: sort by (value descending, index ascending)
        key = flat.double() * 1e7 - torch.arange(n, device="cuda").double() * 1e-3
        ref = key.argsort(dim=1, descending=True)[:, :k]
        assert torch.equal(t, ref)
        assert torch.equal(t[1], torch.arange(k, device="cuda"))   # the constant image: the first k indices


@pytest.mark.parametrize("case", [(3, [1000, 1000], 200, 0.05, False), (3, [1000, 700], 200, 0.05, True),
                                  (5, [300, 2048, 17], 50, 0.3, True), (2, [64, 0], 100, 0.05, False),
                                  (3, [500, 500], 0, 0.05, True), (3, [40, 30], 200, 0.99, False)])
def test_det_postprocess_kernel(hip, case):
    """mmt_det_postprocess (PostProcessor.filter_results on the device: threshold, stable descending sort, NMS, ascending-row
    order, DETECTIONS_PER_IMG cut with ties kept) against the tensor formulation it replaces -- same boxes, scores and
    labels in the same order.  Cases: the bench shape, ragged images, ties in the scores (quantised), an empty image, no
    cut (D = 0), nothing above the threshold."""
    from maskrcnn_benchmark.modeling.roi_heads.box_head.box_head import PostProcessor
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    nc, per, D, thr, ties = case
    g = torch.Generator().manual_seed(1000 + nc + sum(per))
    R = sum(per)
    logits = torch.randn(R, nc, generator=g) * 2.0
    if ties:
        logits = (logits * 2).round() / 2            # many equal probabilities
    ctr = torch.rand(R, 2, generator=g) * 400 + 50
    wh = torch.rand(R, 2, generator=g) * 120 + 8
    props = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    reg = torch.randn(R, nc * 4, generator=g) * 0.5
    boxes, lo = [], 0
    for n in per:
        boxes.append(BoxList(props[lo:lo + n].cuda(), (512, 512), "xyxy"))
        lo += n
    pp = PostProcessor(thr, 0.5, D).cuda()
    x = (logits.cuda(), reg.cuda())
    got = pp(x, boxes)
    # the tensor formulation on the same softmax / decode (tests/tensor_formulations.py::det_filter_results)
    from maskrcnn_benchmark import _hip as H
    from maskrcnn_benchmark.utils.miscellaneous import dev_const
    prob = torch.softmax(x[0], -1)
    offs = [0]
    for n in per:
        offs.append(offs[-1] + n)
    dec = H.box_decode(x[1].reshape(sum(per), -1), torch.cat([b.bbox for b in boxes], 0), pp.box_coder.weights, pp.box_coder.bbox_xform_clip,
                       dev_const(offs, torch.int32, prob.device),
                       dev_const([[b.size[0] - 1, b.size[1] - 1] for b in boxes], torch.float32, prob.device))
    want = tf.det_filter_results(prob, dec, per, [b.size for b in boxes], thr, 0.5, D, H.nms_batched)
    assert len(got) == len(want) == len(per)
    for a, (bb, sc, lb) in zip(got, want):
        assert a.bbox.shape == bb.shape
        assert torch.equal(a.bbox, bb)
        assert torch.equal(a.get_field("scores"), sc)
        assert torch.equal(a.get_field("labels"), lb)
    if D > 0 and not ties:
        assert all(len(a) <= D for a in got)
