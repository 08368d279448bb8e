"""Kernel-level parity: every libmmtpsm.so entry point against the oracle / golden vectors.
Runs on the MI355X box only (-m gpu); calls go through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import gold, T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from maskrcnn_benchmark import _hip
    _hip.lib()
    assert torch.cuda.is_available()
    return _hip


@pytest.fixture(params=["f16x2", "bf16x3"])
def arith(request, hip):
    """mode 3 in both of its arithmetics: the default two-term fp16 split (3 products) and the three-term bf16 split
    (6 products) that it falls back to per tensor -- same tolerances"""
    hip.set_f16x2(request.param == "f16x2")
    yield request.param
    hip.set_f16x2(None)


def cl(x):  # NCHW cpu tensor -> NHWC-dense cuda tensor
    return x.cuda().contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------ ROIAlign
def test_roi_align_forward_golden(hip):
    g = gold("roi_align")
    for i in range(int(g["n"])):
        sc, ph, pw, sr = g["p%d" % i]
        x, r = T(g["x%d" % i]), T(g["r%d" % i])
        lv = torch.zeros(r.shape[0], dtype=torch.int32)
        y = hip.roi_align_forward([cl(x)], [float(sc)], r.cuda(), lv.cuda(), int(ph), int(pw), int(sr))
        ref = g["y%d" % i]
        assert tuple(y.shape) == ref.shape
        np.testing.assert_array_equal(y.cpu().numpy(), ref)  # bit-exact: same op order, no FMA contraction


def test_roi_align_fpn_fused_and_backward(hip):
    from oracle import native
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, 32, 64 >> l, 80 >> l, generator=g) for l in range(4)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    K = 300
    xy = torch.rand(K, 2, generator=g) * torch.tensor([300., 240.]) - 10
    wh = torch.rand(K, 2, generator=g) * 150 + 1
    rois = torch.cat([(torch.arange(K) % 2).float()[:, None], xy, xy + wh], 1)
    lv = (torch.rand(K, generator=g) * 4).long().clamp(max=3)
    for res in (7, 14):
        y = hip.roi_align_forward([cl(f) for f in feats], scales, rois.cuda(), lv.cuda().int(), res, res, 2)
        ref = torch.zeros(K, 32, res, res)
        for l in range(4):
            idx = (lv == l).nonzero().squeeze(1)
            ref[idx] = native.roi_align_forward(feats[l], rois[idx], scales[l], res, res, 2)
        np.testing.assert_array_equal(y.cpu().numpy(), ref.numpy())
        go = torch.randn(K, 32, res, res, generator=g)
        grads = hip.roi_align_backward(cl(go), [f.shape for f in feats], scales, rois.cuda(), lv.cuda().int(), res, res, 2)
        for l in range(4):
            idx = (lv == l).nonzero().squeeze(1)
            gr = native.roi_align_backward(go[idx], rois[idx], scales[l], res, res, *feats[l].shape, 2)
            # fp32 atomics: the order of a texel's ~100 contributions changes from run to run, and where they cancel the error is
            # eps x sum |terms|, not eps x |result| (seen once in ~6 runs: 1.24e-5 on a 1e-2 result with the old absolute 1e-5):
            # the absolute part of the tolerance goes with the magnitude of the gradient map
            np.testing.assert_allclose(grads[l].cpu().numpy(), gr.numpy(), rtol=1e-4, atol=5e-6 * max(1.0, float(gr.abs().max())))


@pytest.mark.parametrize("C,sizes", [(64, [(50, 67), (25, 34), (13, 17), (7, 9)]), (256, [(64, 80), (32, 40), (16, 20), (8, 10)]),
                                     (128, [(40, 40)])])
def test_roi_align_backward_tile_gather(hip, C, sizes, monkeypatch):
    """the backward without atomics (mmt_roi_align_backward_dense: one block per 8 x 8 pixel tile gathers the bins that reach it)
    against the oracle's restatement of cuda/ROIAlign_cuda.cu:177-254 and against the scatter kernel; every element of every
    level written (buffers pre-filled with NaN), repeatable bit for bit; ROIs hanging over every border, sub-pixel ROIs,
    maps that are no multiple of the tile, ROIs of other images / levels"""
    import ctypes
    from oracle import native
    H_ = hip
    g = torch.Generator().manual_seed(C + len(sizes))
    L = len(sizes)
    N = 2
    scales = [0.25 / (1 << l) for l in range(L)]
    K = 260
    W0, H0 = sizes[0][1] * 4, sizes[0][0] * 4
    xy = torch.rand(K, 2, generator=g) * torch.tensor([W0 + 40., H0 + 40.]) - 30
    wh = torch.rand(K, 2, generator=g) * 150 + 1
    wh[::7] = torch.rand(wh[::7].shape, generator=g) * 3      # sub-pixel at every level
    wh[3::11] = 400.                                          # larger than the map
    rois = torch.cat([(torch.arange(K) % N).float()[:, None], xy, xy + wh], 1)
    lv = (torch.rand(K, generator=g) * L).long().clamp(max=L - 1)
    shapes = [(N, C, h, w) for h, w in sizes]
    for res in (7, 14):
        go = torch.randn(K, C, res, res, generator=g)
        monkeypatch.setenv("MMT_ROI_BWD_DENSE", "1")
        dense = H_.roi_align_backward(cl(go), shapes, scales, rois.cuda(), lv.cuda().int(), res, res, 2)
        # really the tile kernel, on buffers that held NaN: called once more through the C ABI
        bufs = [torch.full((N, h, w, C), float("nan"), device="cuda") for h, w in sizes]
        p = H_._pyramid([b.permute(0, 3, 1, 2) for b in bufs], scales, bufs)
        r_d, l_d, g_d = rois.cuda(), lv.cuda().int(), cl(go)
        rc = H_.lib().mmt_roi_align_backward_dense(ctypes.byref(p), r_d.data_ptr(), l_d.data_ptr(), K, res, res, 2,
                                                   g_d.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        for l in range(L):
            assert torch.equal(bufs[l].permute(0, 3, 1, 2), dense[l])     # no NaN left, repeatable
        monkeypatch.setenv("MMT_ROI_BWD_DENSE", "0")
        scatter = H_.roi_align_backward(cl(go), shapes, scales, rois.cuda(), lv.cuda().int(), res, res, 2)
        monkeypatch.delenv("MMT_ROI_BWD_DENSE")
        for l in range(L):
            idx = (lv == l).nonzero().squeeze(1)
            gr = native.roi_align_backward(go[idx], rois[idx], scales[l], res, res, *shapes[l], 2)
            tol = 5e-6 * max(1.0, float(gr.abs().max()))
            np.testing.assert_allclose(dense[l].cpu().numpy(), gr.numpy(), rtol=1e-4, atol=tol)
            np.testing.assert_allclose(dense[l].cpu().numpy(), scatter[l].cpu().numpy(), rtol=1e-4, atol=tol)
    # no ROI at all: zeros everywhere
    monkeypatch.setenv("MMT_ROI_BWD_DENSE", "1")
    z = H_.roi_align_backward(cl(torch.zeros(0, C, 7, 7)), shapes, scales, torch.zeros(0, 5).cuda(), torch.zeros(0).int().cuda(), 7, 7, 2)
    assert all(float(t.abs().max()) == 0.0 for t in z)


# ------------------------------------------------------------------------------------------ NMS
def _nms_via_hip(hip, boxes, scores, thr):
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order].cuda()
    n = b.shape[0]
    seg = torch.tensor([0, n], dtype=torch.int32).cuda()
    keep, cnt = hip.nms_batched(b, seg, max(n, 1), thr)
    k = keep[0, :int(cnt[0])].cpu().long()
    return torch.sort(order[k])[0]


def test_nms_golden(hip):
    g = gold("nms")
    for i in range(6):
        keep = _nms_via_hip(hip, T(g["b%d" % i]), T(g["s%d" % i]), float(g["t%d" % i]))
        assert keep.tolist() == g["k%d" % i].tolist(), i


def test_nms_batched_segments(hip):
    from oracle import native
    g = torch.Generator().manual_seed(9)
    segs, boxes = [0], []
    refs = []
    for n in (2000, 1, 777, 64, 65, 1999, 128, 0, 300, 2000):
        xy = torch.rand(n, 2, generator=g) * 600
        wh = torch.rand(n, 2, generator=g) * 200 + 1
        b = torch.cat([xy, xy + wh], 1)
        s = torch.sort(torch.rand(n, generator=g), descending=True)[0]
        boxes.append(b)
        segs.append(segs[-1] + n)
        refs.append(native.nms(b, s, 0.7))
    keep, cnt = hip.nms_batched(torch.cat(boxes).cuda(), torch.tensor(segs, dtype=torch.int32).cuda(), 2000, 0.7)
    for i, r in enumerate(refs):
        assert keep[i, :int(cnt[i])].cpu().tolist() == r.tolist(), i


# ------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 40, 48, 256, 1, 1, 0),
    (2, 256, 40, 48, 128, 1, 2, 0),
    (2, 128, 20, 24, 128, 3, 1, 1),
    (1, 256, 64, 64, 256, 3, 1, 1),
    (2, 4, 64, 80, 64, 7, 2, 3),
    (2, 256, 16, 20, 15, 1, 1, 0),
    (3, 96, 9, 11, 40, 3, 1, 1),
    (1, 12544, 1, 1, 1024, 1, 1, 0),
    (2, 16, 130, 130, 200, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_epilogues(hip, arith, case):
    N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    if Cin == 12544:
        N = 300
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, s, p)
    res = torch.randn(ref.shape, generator=g)
    y = hip.conv_forward(cl(x), cl(w), stride=s, pad=p)
    scl = ref.abs().max().item()
    assert (y.cpu().double() - ref).abs().max().item() < 2e-5 * max(scl, 1.0)
    y2 = hip.conv_forward(cl(x), cl(w), scale.cuda(), shift.cuda(), s, p, relu=True, res=cl(res), res_mode=1)
    ref2 = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    assert (y2.cpu().double() - ref2).abs().max().item() < 2e-5 * max(scl, 1.0)


def test_conv_fpn_residual_modes_and_scatter(hip, arith):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 16, 24, generator=g)
    w = torch.randn(32, 64, 1, 1, generator=g) / 8
    b = torch.randn(32, generator=g)
    top = torch.randn(2, 32, 8, 12, generator=g)
    y = hip.conv_forward(cl(x), cl(w), None, b.cuda(), res=cl(top), res_mode=2)
    ref = F.conv2d(x, w, b) + F.interpolate(top, scale_factor=2, mode="nearest")
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    fine = torch.randn(2, 32, 32, 48, generator=g)
    y = hip.conv_forward(cl(x), cl(w), res=cl(fine), res_mode=3)
    ref = F.conv2d(x, w) + F.avg_pool2d(fine, 2) * 4
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # strided scatter + relu-mask (data gradient of a stride-2 1x1 conv into a 33x49 input)
    msk = torch.randn(2, 32, 33, 49, generator=g)
    xs = x[:, :, :, :24]
    xs = F.pad(xs, (0, 1, 0, 1))[:, :, :17, :25]
    y = hip.conv_forward(cl(xs), cl(w), out_stride=2, out_hw=(33, 49), mask=cl(msk), mask_scale=2.0)
    ref = torch.zeros(2, 32, 33, 49)
    ref[:, :, ::2, ::2] = F.conv2d(xs, w)
    ref = ref * (msk > 0).float() * 2.0
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    mul = torch.rand(2, 32, 16, 24, generator=g)
    y = hip.conv_forward(cl(x), cl(w), relu=True, mul=cl(mul))
    np.testing.assert_allclose(y.cpu().numpy(), (F.relu(F.conv2d(x, w)) * mul).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", [(2, 128, 20, 24, 128, 3, 1, 1), (2, 256, 24, 24, 128, 1, 2, 0),
                                  (2, 256, 16, 20, 15, 1, 1, 0), (3, 32, 9, 11, 40, 3, 1, 1),
                                  (64, 1024, 1, 1, 12, 1, 1, 0)])
def test_conv_wgrad_and_dgrad(hip, arith, case):
    N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_()
    bnscale = torch.rand(Cout, generator=g) + 0.5
    y = F.conv2d(x, w, None, s, p) * bnscale.view(1, -1, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw = torch.zeros(Cout, Cin, k, k).cuda().contiguous(memory_format=torch.channels_last)
    dw = hip.nhwc(dw)
    db = torch.zeros(Cout).cuda()
    hip.conv_wgrad(cl(x.detach()), cl(dy), (Cout, Cin, k, k), s, p, dw, rowscale=bnscale.cuda(), dbias=db)
    tol = 3e-5 * max(1.0, w.grad.abs().max().item()) * (N * H * W) ** 0.5
    assert (dw.cpu() - w.grad).abs().max().item() < tol
    np.testing.assert_allclose(db.cpu().numpy(), dy.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    if s == 1:
        wd = hip.weight_flip_transpose(cl(w.detach()), bnscale.cuda())
        dx = hip.conv_forward(cl(dy), wd, stride=1, pad=k - 1 - p)
        assert (dx.cpu() - x.grad).abs().max().item() < 3e-5 * max(1.0, x.grad.abs().max().item()) * 10
    else:
        wd = hip.weight_flip_transpose(cl(w.detach()), bnscale.cuda())
        dx = hip.conv_forward(cl(dy), wd, out_stride=s, out_hw=(H, W))
        assert (dx.cpu() - x.grad).abs().max().item() < 3e-4


def test_maxpool(hip):
    x = torch.randn(2, 64, 33, 47)
    y = hip.maxpool3x3s2(cl(x))
    np.testing.assert_array_equal(y.cpu().numpy(), F.max_pool2d(x, 3, 2, 1).numpy())


# ------------------------------------------------------------------------------------------ losses
def test_mask_bce(hip):
    g = torch.Generator().manual_seed(2)
    P = 37
    logits = (torch.randn(P, 3, 28, 28, generator=g) * 3).requires_grad_()
    labels = (torch.rand(P, generator=g) * 2).long() + 1
    tgt = (torch.rand(P, 28, 28, generator=g) > 0.5).float()
    ref = F.binary_cross_entropy_with_logits(logits[torch.arange(P), labels], tgt)
    ref.backward()
    loss, grad = hip.mask_bce(cl(logits.detach()), labels.cuda(), tgt.cuda(), 1.0)
    assert loss.item() == pytest.approx(ref.item(), rel=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), logits.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_mgd_golden_and_grad(hip):
    from oracle import model as om
    g = gold("mt_losses")
    for name, nl in (("mgd0", 2), ("mgd1", 5)):
        tp = [[T(g["%s_t%d_%d" % (name, i, l)]) for l in range(nl)] for i in range(4)]
        sp = [T(g["%s_s_%d" % (name, l)]).clone().requires_grad_() for l in range(nl)]
        masks = T(g[name + "_m"])
        ref = om.fg_hint_loss(tp, [sp], [m for m in masks])
        ref.backward()
        seg = masks.int().cuda()
        terms = []
        accs = []
        for l in range(nl):
            N, C, H, W = sp[l].shape
            m = hip.mask_pool(seg, H, W)
            acc = hip.mgd_level_forward(cl(sp[l].detach()), [cl(tp[i][l]) for i in range(4)], [i % 2 == 1 for i in range(4)], m)
            den = acc[4] * C + 1e-7
            terms.append(acc[:4] / den)
            accs.append((m, den))
        loss = torch.cat(terms).mean()
        assert loss.item() == pytest.approx(float(g[name]), rel=2e-5)
        for l in range(nl):
            m, den = accs[l]
            coef = torch.full((4,), 1.0 / (4 * nl), device="cuda") / den
            gs = hip.mgd_level_backward(cl(sp[l].detach()), [cl(tp[i][l]) for i in range(4)],
                                        [i % 2 == 1 for i in range(4)], m, coef)
            np.testing.assert_allclose(gs.cpu().numpy(), sp[l].grad.numpy(), rtol=1e-4, atol=1e-7)


def test_psm_golden_and_grad(hip):
    from oracle import model as om
    g = gold("mt_losses")
    for case, typ in enumerate(("bce", "bce", "kl", "mse")):
        cfg = om.default_cfg(mt_cls_loss_type=typ)
        t = T(g["psm%d_t" % case])
        s = T(g["psm%d_s" % case]).clone().requires_grad_()
        labels = T(g["psm%d_labels" % case])
        ref = om.psm_loss(cfg, [s], [x for x in t], labels)
        ref.backward()
        v = hip.psm_variance(t.cuda(), use_softmax=(typ == "bce"))
        vref = torch.std(torch.stack([F.softmax(x, 1) if typ == "bce" else x for x in t]), dim=0).sum(-1)
        np.testing.assert_allclose(v.cpu().numpy(), vref.numpy(), rtol=1e-4, atol=1e-6)
        pos, neg = labels > 0, labels == 0
        vn = torch.where(neg, vref, torch.full_like(vref, -1.0))
        order = torch.argsort(vn, descending=True, stable=True)
        nkeep = min(int(neg.sum()), int(pos.sum()) // 2)
        roww = pos.float()
        w_neg = 1.5 if typ == "bce" else 1.0
        roww[order[:nkeep]] = w_neg
        S = int(pos.sum()) + nkeep
        kind = {"bce": 0, "kl": 1, "mse": 2}[typ]
        rl, rg = hip.psm_rows(t.cuda(), s.detach().cuda(), roww.cuda(), 0.5, 1, kind)
        norm = 1.0 / (S * 3)  # NC = 3: 'ce' .mean(0).sum()/3 and the element mean of 'kl' / 'mse' coincide
        assert (rl.sum() * norm).item() == pytest.approx(float(g["psm%d" % case]), rel=2e-5)
        np.testing.assert_allclose((rg * norm).cpu().numpy(), s.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_evaluate_psm_host_function_golden(hip):
    """FastRCNNLossComputation.evaluatePSM as the model calls it (device ranking of the hard negatives included) against
    the reference's values for every CLS_LOSS_TYPE and for RANK_FILTER 0 -- the yacs default, where 'bce' feeds the
    mean of the per-view softmax probabilities through softmax once more (box_head/loss.py:164-173,229,281)"""
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.roi_heads.box_head.box_head import make_roi_box_loss_evaluator
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    g = gold("mt_losses")
    for case, (typ, rf) in enumerate((("bce", 0.2), ("bce", 0.2), ("kl", 0.2), ("mse", 0.2), ("bce", 0.0))):
        cfg = make_default_cfg()
        cfg.merge_from_list(["MT.CLS_LOSS_TYPE", typ, "MT.RANK_FILTER", rf])
        ev = make_roi_box_loss_evaluator(cfg)
        labels = T(g["psm%d_labels" % case]).cuda()
        bl = BoxList(torch.zeros(len(labels), 4).cuda(), (10, 10), "xyxy")
        bl.add_field("labels", labels)
        s = T(g["psm%d_s" % case]).cuda().requires_grad_()
        v = ev.evaluatePSM([s], [x for x in T(g["psm%d_t" % case]).cuda()], [bl])
        assert v.item() == pytest.approx(float(g["psm%d" % case]), rel=2e-5), (case, typ, rf)
        v.backward()
        assert torch.isfinite(s.grad).all() and s.grad.abs().sum().item() > 0


def test_ema_and_sgd(hip):
    g = gold("mt_losses")
    from oracle import model as om
    cfg = om.default_cfg()
    t = torch.zeros(8).cuda()
    s0 = torch.zeros(8)
    s0[:5] = torch.arange(5).float()
    for it in range(21):
        hip.ema_update(t, (s0 * (1 + 0.1 * it)).cuda(), om.ema_alpha(cfg, it))
        np.testing.assert_array_equal(t[:5].cpu().numpy(), g["ema_trace"][it])
    n = 1000003
    gen = torch.Generator().manual_seed(1)
    a, b = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    ta = a.clone()
    ta.mul_(0.99).add_(b, alpha=1 - 0.99)
    ca = a.cuda()
    hip.ema_update(ca, b.cuda(), 0.99)
    np.testing.assert_allclose(ca.cpu().numpy(), ta.numpy(), rtol=1e-6, atol=1e-7)
    # SGD vs torch.optim.SGD over 3 steps
    p = torch.nn.Parameter(a.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    cp, buf = a.clone().cuda(), torch.zeros(n).cuda()
    for step in range(3):
        gr = torch.randn(n, generator=gen)
        p.grad = gr.clone()
        opt.step()
        hip.sgd_momentum(cp, gr.cuda(), buf, 0.01, 1e-4, 0.9, step == 0)
    np.testing.assert_allclose(cp.cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ masks
def test_paste_masks_golden(hip):
    g = gold("masks")
    probs, boxes, ref = T(g["paste_masks"]), T(g["paste_boxes"]), g["paste_out"]
    logits = torch.log(probs / (1 - probs))  # sigmoid^-1; the kernel applies sigmoid itself
    D = probs.shape[0]
    lg = torch.zeros(D, 3, 28, 28)
    lg[:, 1] = logits
    seg = hip.paste_masks(cl(lg), torch.ones(D).int().cuda(), boxes.cuda(), torch.zeros(D).int().cuda(), 1, 128, 150, 0.5)
    want = ref.astype(np.int32).sum(0)
    got = seg[0].cpu().numpy()
    mism = (got != want).sum()
    # logit->sigmoid round trip may move a probability across 0.5 by one ulp; allow a handful of pixels
    assert mism <= 3, mism


def test_polygon_targets_golden(hip):
    g = gold("masks")
    lens = g["proj_polylens"]
    poly_off = np.concatenate([[0], np.cumsum(lens // 2)]).astype(np.int32)
    ends = np.cumsum(g["proj_npoly"])
    roi_poly = np.stack([ends - g["proj_npoly"], ends], 1).astype(np.int32)
    out, ovf = hip.polygon_targets(T(g["proj_polys"]).cuda(), T(poly_off).cuda(), T(roi_poly).cuda(),
                                   T(g["proj_boxes"]).cuda(), 28)
    assert int(ovf) == 0
    np.testing.assert_array_equal(out.cpu().numpy(), g["proj_out"])


# ------------------------------------------------------------------------------------------ conv arithmetic modes
@pytest.fixture
def restore_mode(hip):
    m = hip.get_conv_precision()
    yield
    hip.set_conv_precision(m)


MODE_TOL = {0: 1e-5, 3: 1e-5, 2: 6e-5, 1: 2e-2}  # max |err| / max |ref| against fp64


@pytest.mark.parametrize("case", [(2, 64, 24, 24, 96, 3, 1, 1), (3, 128, 30, 30, 160, 1, 2, 0), (1, 48, 17, 23, 64, 3, 1, 1),
                                  (2, 256, 64, 64, 256, 3, 1, 1), (2, 16, 40, 40, 64, 1, 1, 0), (8, 32, 64, 64, 64, 1, 1, 0),
                                  (300, 1024, 1, 1, 1024, 1, 1, 0), (2, 256, 32, 32, 256, 3, 1, 1), (200, 4096, 1, 1, 256, 1, 1, 0)])
def test_conv_arithmetic_modes_forward(hip, restore_mode, arith, case):
    """every arithmetic mode of mmt_conv_forward against fp64, with and without pre-packed weight planes; the 3-term
    split (default) must be as accurate as the fp32-input MFMA"""
    N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x, w = cl(torch.randn(N, Cin, H, W, generator=g)), cl(torch.randn(Cout, Cin, k, k, generator=g) * 0.05)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    ref = F.relu(F.conv2d(x.double(), w.double(), None, s, p) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    scl = ref.abs().max().item()
    err = {}
    for mode in (0, 3, 2, 1):
        hip.set_conv_precision(mode)
        y = hip.conv_forward(x, w, sc, sh, s, p, relu=True)
        err[mode] = (y.double() - ref).abs().max().item() / scl
        assert err[mode] < MODE_TOL[mode], (mode, err[mode])
    assert err[3] <= 2.0 * err[0] + 2e-7, err
    # tiles hanging over M and Cout, zero-filled halo, K = 16 (a single step) are all in the cases above; the last two
    # (few tiles, K >= 2048) take the split-K form of the DMA kernel (partials summed in a fixed order: repeatable)
    hip.set_conv_precision(3)
    assert torch.equal(hip.conv_forward(x, w, sc, sh, s, p, relu=True), hip.conv_forward(x, w, sc, sh, s, p, relu=True))


@pytest.mark.parametrize("case", [(2, 64, 24, 24, 96, 3, 1, 1), (3, 128, 30, 30, 160, 1, 2, 0), (1, 48, 17, 23, 64, 3, 1, 1),
                                  (1000, 256, 1, 1, 512, 1, 1, 0), (2, 20, 12, 12, 36, 3, 1, 1), (20, 256, 14, 14, 256, 3, 1, 1),
                                  (2, 32, 40, 40, 64, 3, 2, 1), (2, 32, 40, 36, 64, 3, 2, 1), (3, 64, 16, 12, 128, 3, 1, 1),
                                  (2, 64, 32, 32, 15, 1, 1, 0), (600, 1024, 1, 1, 15, 1, 1, 0), (2, 32, 20, 20, 30, 3, 1, 1)])
def test_conv_arithmetic_modes_wgrad(hip, restore_mode, arith, case):
    """the three pixel-decode variants of the pipelined kernel are all here: Wo % 4 == 0 (one carried position per
    thread, strides 1 and 2), other Wo >= 8 (one per pixel), and 1 x 1 / tiny maps (divisions); Cout % 4 != 0 (the
    15-channel predictors) takes the element-wise dy loads"""
    N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = cl(torch.randn(N, Cin, H, W, generator=g))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = cl(torch.randn(N, Cout, Ho, Wo, generator=g))
    rs = (torch.rand(Cout, generator=g) + 0.5).cuda()
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), stride=s, padding=p) * rs.double().view(-1, 1, 1, 1)
    scl = ref.abs().max().item()
    err = {}
    for mode in (0, 3, 2, 1):
        hip.set_conv_precision(mode)
        dw = hip.nhwc(torch.zeros(Cout, Cin, k, k, device="cuda").contiguous(memory_format=torch.channels_last))
        hip.conv_wgrad(x, dy, (Cout, Cin, k, k), s, p, dw, rowscale=rs)
        hip.conv_wgrad(x, dy, (Cout, Cin, k, k), s, p, dw, rowscale=rs)  # accumulates
        err[mode] = (dw.double() - 2 * ref).abs().max().item() / (2 * scl)
        assert err[mode] < MODE_TOL[mode], (mode, err[mode])
    assert err[3] <= 2.0 * err[0] + 2e-7, err


@pytest.mark.parametrize("case", [(4, 64, 128, 128, 256), (1, 64, 257, 259, 96), (1, 128, 300, 300, 200), (2, 128, 192, 192, 512), (2, 128, 128, 128, 512)])
@pytest.mark.parametrize("residual", [False, True])
def test_conv1x1_rows_kernel(hip, restore_mode, case, residual):
    """1x1 layers with K = 64 / 128 and >= 64k rows run on conv1x1_rows_kernel (one block per 128 rows, all Cout
    panels): bit-identical to the tiled kernel (MMT_ROWS=0), fp32-grade against fp64; ragged M and Cout included"""
    import os
    hip.set_f16x2(False)   # this kernel and the tiled kernel it is compared with: the 3-term bf16 split (bit-identical there)
    N, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x, w = cl(torch.randn(N, Cin, H, W, generator=g)), cl(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    res = cl(torch.randn(N, Cout, H, W, generator=g)) if residual else None
    ref = F.conv2d(x.double(), w.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    ref = F.relu(ref + res.double()) if residual else F.relu(ref)
    old = os.environ.get("MMT_ROWS")
    try:
        for mode in (3, 1):
            hip.set_conv_precision(mode)
            os.environ["MMT_ROWS"] = "1"
            y1 = hip.conv_forward(x, w, sc, sh, 1, 0, relu=True, res=res, res_mode=1 if residual else 0)
            os.environ["MMT_ROWS"] = "0"
            y0 = hip.conv_forward(x, w, sc, sh, 1, 0, relu=True, res=res, res_mode=1 if residual else 0)
            assert torch.equal(y0, y1), mode
            assert (y1.double() - ref).abs().max().item() < MODE_TOL[mode] * ref.abs().max().item()
    finally:
        if old is None:
            os.environ.pop("MMT_ROWS", None)
        else:
            os.environ["MMT_ROWS"] = old


def test_packed_weight_planes_bookkeeping(hip, restore_mode, arith):
    """engine/flat.py keeps one packed copy of all weight matrices; a convolution must never use a stale one"""
    from torch import nn
    from maskrcnn_benchmark.layers import Conv2d
    from maskrcnn_benchmark.engine.flat import flatten_model
    hip.set_conv_precision(3)
    torch.manual_seed(3)
    m = nn.Sequential(Conv2d(32, 64, 3, 1, 1), Conv2d(64, 48, 1, 1, 0)).cuda()
    x = cl(torch.randn(2, 32, 20, 20))

    def fwd():
        with torch.no_grad():
            return m[1](m[0](x, relu=True))

    def ref():
        with torch.no_grad():
            return F.conv2d(F.relu(F.conv2d(x.double(), m[0].weight.double(), m[0].bias.double(), 1, 1)),
                            m[1].weight.double(), m[1].bias.double())

    def close(y):
        r = ref()
        return (y.double() - r).abs().max().item() < 1e-5 * r.abs().max().item()

    y_percall = fwd()  # not flattened: packed per call
    flat = flatten_model(m)
    w0 = m[0].weight
    assert hip.PLANES[w0.data_ptr()][0]() is flat and flat.planes is not None
    assert torch.equal(fwd(), y_percall) and close(y_percall)  # same arithmetic through the cached planes
    # (a) raw-pointer update + refresh (what FlatSGD.step / update_teacher do)
    flat.data.mul_(1.5)
    flat.refresh_planes()
    assert close(fwd())
    # (b) in-place update through torch WITHOUT a refresh: the version counter moved -> per-call packing, still right
    with torch.no_grad():
        m[0].weight.mul_(-0.5)
    assert close(fwd())
    # (c) a mode switch invalidates planes packed before it (steps taken in mode 0 do not refresh them)
    hip.set_conv_precision(0)
    flat.data.mul_(2.0)
    flat.refresh_planes()  # no-op in mode 0
    hip.set_conv_precision(3)
    assert close(fwd())
    flat.refresh_planes()
    assert close(fwd())
    # (d) the data-gradient planes are cached per parameter generation (H.FLIPPED): reused inside a generation, rebuilt
    # after the parameters moved
    h = cl(torch.randn(2, 64, 20, 20))

    def dgrad():
        xx = h.clone().requires_grad_(True)
        m[1](xx).sum().backward()
        return xx.grad

    def dgrad_ref():
        xx = h.double().clone().requires_grad_(True)
        F.conv2d(xx, m[1].weight.double(), m[1].bias.double()).sum().backward()
        return xx.grad

    g1 = dgrad()
    key1 = hip.FLIPPED[m[1].weight.data_ptr()][0]
    g2 = dgrad()
    assert hip.FLIPPED[m[1].weight.data_ptr()][0] == key1 and torch.equal(g1, g2)
    flat.data.mul_(0.5)
    flat.refresh_planes()
    g3, r3 = dgrad(), dgrad_ref()
    assert hip.FLIPPED[m[1].weight.data_ptr()][0] != key1
    assert (g3.double() - r3).abs().max().item() < 1e-5 * r3.abs().max().item()


def test_batch_pack_maxima_are_the_matrices_maxima(hip, restore_mode):
    """mmt_pack_weights_f16 / mmt_pack_weights_flipped_f16 derive every matrix's power-of-two scale from max |w| (max |w * scale[co]|
    for the data-gradient form), found by a reduction launch that reads the matrices in memory order (round 6: the maximum does not
    care about the packing order): exact, for matrices whose row count is not a multiple of 32, several tap counts, with and
    without a row scale; the planes built on it reproduce the layer"""
    from torch import nn
    from maskrcnn_benchmark.layers import Conv2d
    from maskrcnn_benchmark.engine.flat import flatten_model
    hip.set_conv_precision(3)
    hip.set_f16x2(True)
    try:
        torch.manual_seed(11)
        m = nn.Sequential(Conv2d(64, 80, 3, 1, 1), Conv2d(80, 48, 1, 1, 0), Conv2d(48, 144, 3, 1, 1), Conv2d(144, 64, 1, 1, 0)).cuda()
        with torch.no_grad():
            for i, l in enumerate(m):
                l.weight.mul_(10.0 ** (i - 2))          # maxima four decades apart: a mixed-up matrix would show
                l.weight[-1, -1, -1, -1] = 3.0 * l.weight.abs().max()   # ... and the maximum in the last element of the matrix in memory
        flat = flatten_model(m)
        x = cl(torch.randn(2, 64, 24, 24)).requires_grad_(True)
        h = x
        for l in m:
            h = l(h)
        h.sum().backward()                               # registers the data-gradient forms with the flat buffer
        flat.refresh_planes()
        torch.cuda.synchronize()
        mats = [p for _, p in flat._named if p.dim() >= 2 and p.shape[0] > 32 and (p.numel() // p.shape[0]) % 16 == 0]
        assert flat.stat16 is not None and flat.stat16.shape[0] == len(mats) == 4
        for d, p in enumerate(mats):
            assert flat.stat16[d, 0].item() == p.detach().abs().max().item(), d
        t = flat.__dict__.get("_flip16_table")
        assert t is not None and len(t[3]) >= 3
        for w, scale, planes, dims in flat._flip16_entries.values():
            want = (w.detach() * (scale.view(-1, 1, 1, 1) if scale is not None else 1.0)).abs().max().item()
            assert t[4][t[3][w.data_ptr()], 0].item() == want, dims
        # the planes reproduce the layers (forward and data gradient) after the bulk re-pack
        x2 = cl(torch.randn(2, 64, 24, 24)).requires_grad_(True)
        h = x2
        for l in m:
            h = l(h)
        h.sum().backward()
        xr = x2.detach().double().requires_grad_(True)
        hr = xr
        for l in m:
            hr = F.conv2d(hr, l.weight.double(), l.bias.double(), 1, l.weight.shape[2] // 2)
        hr.sum().backward()
        assert (h.double() - hr).abs().max().item() < 1e-5 * hr.abs().max().item()
        assert (x2.grad.double() - xr.grad).abs().max().item() < 1e-5 * xr.grad.abs().max().item()
    finally:
        hip.set_f16x2(None)


# ------------------------------------------------------------------------------------------ target assignment
@pytest.mark.parametrize("lowq", [False, True])
def test_match_targets_against_tensor_formulation(hip, lowq):
    """mmt_match_targets == boxlist_iou + Matcher + label rules + BoxCoder.encode (bit-exact), several images, ties"""
    from maskrcnn_benchmark.modeling.matcher import Matcher
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.structures.boxlist_ops import box_iou_tensor
    g = torch.Generator().manual_seed(7 + lowq)
    hi, lo = (0.7, 0.3) if lowq else (0.5, 0.5)
    matcher, coder = Matcher(hi, lo, allow_low_quality_matches=lowq), BoxCoder(weights=(10., 10., 5., 5.))

    def boxes(n, scale):
        xy = torch.rand(n, 2, generator=g) * scale
        wh = torch.rand(n, 2, generator=g) * scale * 0.4 + 2
        return torch.cat([xy, xy + wh], 1).round()  # integer coordinates -> exact IoU ties are likely

    def area(b):
        return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)

    gts = [boxes(5, 60), boxes(9, 60), boxes(1, 60)]
    gls = [torch.randint(1, 3, (len(b),), generator=g) for b in gts]
    # per-image candidates (box head form); image 1 contains exact copies of its gts and duplicates (ties)
    cands = [boxes(300, 60), torch.cat([boxes(200, 60), gts[1], gts[1][:3]], 0), boxes(64, 60)]
    coff = torch.tensor([0, 300, 512, 576], dtype=torch.int32).cuda()
    goff = torch.tensor([0, 5, 14, 15], dtype=torch.int32).cuda()
    m, lab, reg = hip.match_targets(torch.cat(cands).cuda(), coff, torch.cat(gts).cuda(), goff, 3, hi, lo, lowq,
                                    gt_labels=torch.cat(gls).cuda(), box_labels=True, weights=coder.weights)
    ms, ls, rs = m.split([300, 212, 64]), lab.split([300, 212, 64]), reg.split([300, 212, 64])
    for i in range(3):
        gt, c = gts[i].cuda(), cands[i].cuda()
        ref = matcher(box_iou_tensor(gt, area(gt), c, area(c)))
        np.testing.assert_array_equal(ms[i].cpu().numpy(), ref.cpu().numpy())
        rl = gls[i].cuda()[ref.clamp(min=0)]
        rl = torch.where(ref == -1, torch.zeros_like(rl), rl)
        rl = torch.where(ref == -2, torch.full_like(rl, -1), rl)
        np.testing.assert_array_equal(ls[i].cpu().numpy(), rl.cpu().numpy())
        np.testing.assert_array_equal(rs[i].cpu().numpy(), coder.encode(gt[ref.clamp(min=0)], c).cpu().numpy())
    # shared anchor grid (RPN form) with visibility
    anc = boxes(5000, 60).cuda()
    vis = (torch.rand(5000, generator=g) > 0.2).cuda()
    coder1 = BoxCoder(weights=(1., 1., 1., 1.))
    m, lab, reg = hip.match_targets(anc, torch.tensor([0, 5000, 10000], dtype=torch.int32).cuda(), torch.cat(gts[:2]).cuda(),
                                    goff[:3].contiguous(), 2, hi, lo, lowq, visible=vis, shared_cand=True, rpn_labels=True,
                                    weights=coder1.weights)
    for i in range(2):
        gt = gts[i].cuda()
        ref = matcher(box_iou_tensor(gt, area(gt), anc, area(anc)))
        rl = (ref >= 0).float()
        rl = torch.where(vis, rl, torch.full_like(rl, -1.0))
        rl = torch.where(ref == -2, torch.full_like(rl, -1.0), rl)
        np.testing.assert_array_equal(lab[i * 5000:(i + 1) * 5000].cpu().numpy(), rl.cpu().numpy())
        np.testing.assert_array_equal(reg[i * 5000:(i + 1) * 5000].cpu().numpy(),
                                      coder1.encode(gt[ref.clamp(min=0)], anc).cpu().numpy())


def test_conv3x3_strip_kernel_planes(hip, restore_mode):
    """conv3x3_strip_kernel (3x3 / stride 1 / pad 1 on pre-split bf16 planes, 256 x 128 tiles on eight waves, one input strip
    per (kh, 16-channel slab) serving the three horizontal taps) against fp64 and against the in-register-split kernel it
    replaces for these shapes: same split, same products, K summed in the order (kh, slab, kw) -- fp32-grade like it.
    Both strip widths, image borders, a Cout that is not a multiple of 128, every epilogue operand."""
    import os
    hip.set_conv_precision(3)
    hip.set_f16x2(False)   # the bf16-plane form of the kernel (the fall-back arithmetic); its fp16 form: tests/test_f16x2_gpu.py
    g = torch.Generator().manual_seed(31)
    taken = 0  # shapes the library ran on the strip kernel: grids of >= 256 tiles (round 4: fp32 few-tile shapes go to the tiled kernel)
    for (N, C, H, W, Co, opts) in ((2, 128, 128, 128, 192, "res"), (8, 256, 64, 64, 256, "relu"), (32, 128, 32, 64, 128, "mask"),
                                   (8, 160, 64, 128, 64, ""), (1, 256, 64, 64, 256, "relu"), (2, 256, 64, 64, 256, "res"),
                                   (2, 128, 128, 128, 128, "relu"), (1, 256, 64, 64, 192, ""), (1, 64, 64, 64, 128, "")):
        x = cl(torch.randn(N, C, H, W, generator=g))
        w = cl(torch.randn(Co, C, 3, 3, generator=g) * 0.05)
        sc, sh = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
        res = cl(torch.randn(N, Co, H, W, generator=g)) if opts == "res" else None
        mask = cl(torch.randn(N, Co, H, W, generator=g)) if opts == "mask" else None
        kw = dict(relu=opts in ("res", "relu"), res=res, res_mode=1 if res is not None else 0, mask=mask, mask_scale=2.0)
        planes = hip.split_planes(x)
        # p0 + p1 + p2 reproduces x to 2^-24 relative (three round-to-nearest bf16 terms)
        rec = planes.float().sum(0).view(N, H, W, C).permute(0, 3, 1, 2)
        assert (rec - x).abs().max().item() <= 2e-7 * x.abs().max().item()
        os.environ["MMT_STRIP"] = "0"
        try:
            y_old = hip.conv_forward(x, w, sc, sh, 1, 1, **kw)
        finally:
            os.environ.pop("MMT_STRIP", None)
        y_new = hip.conv_forward(x, w, sc, sh, 1, 1, x_planes=planes, **kw)
        y_auto = hip.conv_forward(x, w, sc, sh, 1, 1, **kw)          # the wrapper splits by itself when the library wants planes
        ref = F.conv2d(x.double(), w.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        if res is not None:
            ref = ref + res.double()
        if kw["relu"]:
            ref = F.relu(ref)
        if mask is not None:
            ref = torch.where(mask > 0, ref * 2.0, torch.zeros_like(ref))
        scale = ref.abs().max().item()
        e_old, e_new = (y_old.double() - ref).abs().max().item() / scale, (y_new.double() - ref).abs().max().item() / scale
        assert e_new < MODE_TOL[3], (N, C, H, W, Co, e_new)
        assert e_new < 2.0 * e_old + 1e-7, (e_new, e_old)
        assert torch.equal(y_new, y_auto)
        if hip.planes_wanted_3x3(N, C, H, W, Co):
            assert not torch.equal(y_new, y_old)   # it really was the other kernel (another summation order)
            taken += 1
        else:
            assert torch.equal(y_new, y_old)       # not a shape for it: planes ignored, same kernel as before
    assert taken >= 4


def test_box_decode_kernel(hip):
    """mmt_box_decode against BoxCoder.decode + clip_to_image evaluated on the HOST (the reference's CPU arithmetic: true
    divisions by the weights -- ATen's device kernels multiply by the reciprocal instead -- every intermediate rounded to
    fp32): equal up to exp() of the two math libraries, i.e. to the last place of the box size; both weight sets, the
    dw/dh clip active, per-image limits with an empty image; and against the reference's own decode outputs."""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    g = gold("small_ops")
    gen = torch.Generator().manual_seed(17)
    for weights, ncls in (((1.0, 1.0, 1.0, 1.0), 1), ((10.0, 10.0, 5.0, 5.0), 3)):
        coder = BoxCoder(weights)
        R = 3001
        xy = torch.rand(R, 2, generator=gen) * 900
        wh = torch.rand(R, 2, generator=gen) * 200 + 1
        boxes = torch.cat([xy, xy + wh], 1)
        codes = torch.randn(R, 4 * ncls, generator=gen) * (2.0 if ncls == 1 else 8.0)
        ref = coder.decode(codes, boxes)                                            # host
        own = hip.box_decode(codes.cuda(), boxes.cuda(), weights, coder.bbox_xform_clip).cpu()
        scale = ref.abs().max(1)[0].clamp(min=1.0)[:, None]                        # box corners grow with exp(dw) * w
        assert ((own - ref).abs() <= 1e-6 * scale + 1e-4).all(), ((own - ref).abs() / scale).max()
        assert (own == ref).float().mean().item() > 0.5                           # mostly the same bits
        assert (codes[:, 2::4] / weights[2] > coder.bbox_xform_clip).any()        # the clip was exercised
        off = torch.tensor([0, 1000, 1000, R], dtype=torch.int32).cuda()           # an empty image in the middle
        lim = torch.tensor([[999.0, 799.0], [10.0, 10.0], [511.0, 639.0]])
        clipped = hip.box_decode(codes.cuda(), boxes.cuda(), weights, coder.bbox_xform_clip, off, lim.cuda()).cpu()
        parts = []
        for i, (a, b) in enumerate(((0, 1000), (1000, 1000), (1000, R))):
            l4 = torch.tensor([lim[i, 0], lim[i, 1], lim[i, 0], lim[i, 1]])
            parts.append(torch.minimum(own[a:b].reshape(-1, 4).clamp(min=0), l4).reshape(b - a, 4 * ncls))
        assert torch.equal(clipped, torch.cat(parts, 0))                            # the clip itself is exact
    for nm, w in (("10", (10., 10., 5., 5.)), ("1", (1., 1., 1., 1.))):             # the reference's own outputs
        own = hip.box_decode(T(g["codes" + nm]).cuda(), T(g["props"]).cuda(), w, BoxCoder(w).bbox_xform_clip)
        np.testing.assert_allclose(own.cpu().numpy(), g["dec" + nm], rtol=1e-6, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("R,frac", [(300000, 0.002), (5000, 0.2), (1000, 0.0)])
def test_rpn_loss_kernel_vs_tensor_formulation(R, frac):
    """mmt_rpn_loss (two launches) against the tensor formulation of rpn/loss.py:183-194 it replaces: values and both gradients"""
    from maskrcnn_benchmark.layers import fused
    torch.manual_seed(R)
    dev = "cuda"
    obj = (torch.randn(R, device=dev) * 3).requires_grad_(True)
    reg = (torch.randn(R, 4, device=dev) * 0.3).requires_grad_(True)
    regt = torch.randn(R, 4, device=dev) * 0.3
    u = torch.rand(R, device=dev)
    pos, neg = u < frac, (u >= frac) & (u < 3 * frac)
    labels = torch.where(pos, torch.ones_like(u), torch.where(neg, torch.zeros_like(u), -torch.ones_like(u)))
    lo, lb = fused.RPNLossFn.apply(obj, reg, labels, regt, pos, neg, 1.0 / 9)
    (2.0 * lo + 3.0 * lb).backward()
    o2, r2 = obj.detach().clone().requires_grad_(True), reg.detach().clone().requires_grad_(True)
    samp = pos | neg
    n = samp.sum().clamp(min=1).float()
    d = torch.abs(r2 - regt)
    beta = 1.0 / 9
    sl1 = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    wb = (sl1 * pos.float()[:, None]).sum() / n
    wo = (torch.nn.functional.binary_cross_entropy_with_logits(o2, labels.clamp(min=0), reduction="none") * samp.float()).sum() / n
    (2.0 * wo + 3.0 * wb).backward()
    # (block partials meet in fp32 atomics: the order changes from run to run)
    assert lo.item() == pytest.approx(wo.item(), rel=1e-5, abs=1e-9)
    assert lb.item() == pytest.approx(wb.item(), rel=1e-5, abs=1e-9)
    assert torch.allclose(obj.grad, o2.grad, rtol=1e-5, atol=1e-9)
    assert torch.allclose(reg.grad, r2.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("R,NC", [(1024, 3), (7, 3), (2000, 81)])
def test_box_loss_kernel_vs_tensor_formulation(R, NC):
    """mmt_box_loss (one launch) against the tensor formulation of box_head/loss.py:118-162 it replaces"""
    from maskrcnn_benchmark.layers import fused
    torch.manual_seed(R + NC)
    dev = "cuda"
    logits = (torch.randn(R, NC, device=dev) * 2).requires_grad_(True)
    breg = (torch.randn(R, 4 * NC, device=dev) * 0.8).requires_grad_(True)
    labels = torch.randint(0, NC, (R,), device=dev) * (torch.rand(R, device=dev) < 0.4)
    regt = torch.randn(R, 4, device=dev) * 0.8
    lc, lb = fused.BoxLossFn.apply(logits, breg, labels, regt)
    (2.0 * lc + 3.0 * lb).backward()
    l2, b2 = logits.detach().clone().requires_grad_(True), breg.detach().clone().requires_grad_(True)
    wc = torch.nn.functional.cross_entropy(l2, labels)
    idx = (4 * labels.clamp(min=0))[:, None] + torch.arange(4, device=dev)[None, :]
    d = torch.abs(torch.gather(b2, 1, idx) - regt)
    wb = (torch.where(d < 1.0, 0.5 * d * d, d - 0.5) * (labels > 0).float()[:, None]).sum() / R
    (2.0 * wc + 3.0 * wb).backward()
    assert lc.item() == pytest.approx(wc.item(), rel=1e-5)
    assert lb.item() == pytest.approx(wb.item(), rel=1e-5, abs=1e-9)
    assert torch.allclose(logits.grad, l2.grad, rtol=1e-5, atol=1e-9)
    assert torch.allclose(breg.grad, b2.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_mask_head_targets_one_launch_matching_vs_tensor_formulation(synth):
    """MaskRCNNLossComputation.prepare_targets: IoU + Matcher + label lookup of all images through `mmt_match_targets` (one
    launch) against the per-image tensor formulation of mask_head/loss.py:119-149 (boxlist_iou, Matcher, gathers): labels and
    mask targets bit-equal, with proposals that are ground truth (IoU 1), jittered boxes, and boxes below the threshold"""
    from test_model_gpu import _targets_product
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.matcher import Matcher
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.mask_head import MaskRCNNLossComputation
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    dev = torch.device("cuda")
    _, tgs = synth.make_labeled(2, 320, 6, seed=77)
    targets = _targets_product(tgs, dev)
    torch.manual_seed(5)
    props = []
    for t in targets:
        g = t.bbox
        jit = g.repeat(4, 1) + torch.randn(4 * len(g), 4, device=dev) * 6.0
        far = torch.rand(10, 4, device=dev) * 40
        far[:, 2:] += far[:, :2] + 5
        props.append(BoxList(torch.cat([g, jit, far]), t.size, "xyxy"))
    ev = MaskRCNNLossComputation(Matcher(0.5, 0.5, allow_low_quality_matches=False), 28, make_default_cfg())
    lab, mt = ev.prepare_targets(props, targets)
    ev.proposal_matcher = Matcher(0.5, 0.5 - 1e-9, allow_low_quality_matches=False)   # unequal thresholds: the tensor formulation
    lab_t, mt_t = ev.prepare_targets(props, targets)
    assert (lab > 0).any() and (lab == 0).any()
    assert torch.equal(lab, lab_t)
    assert torch.equal(mt, mt_t)


def test_box_loss_on_a_fixed_capacity_batch(hip_lib=None):
    """mmt_box_loss_rows (round 6, SURVEY f-2): rows labelled -1 are padding -- no loss, zero gradient -- and both means run over the
    counted rows: equal to mmt_box_loss on the batch with those rows removed"""
    from maskrcnn_benchmark import _hip as H
    g = torch.Generator().manual_seed(3)
    R, NC = 1024, 3
    logits = torch.randn(R, NC, generator=g).cuda()
    breg = torch.randn(R, 4 * NC, generator=g).cuda()
    labels = torch.randint(0, NC, (R,), generator=g).cuda()
    regt = torch.randn(R, 4, generator=g).cuda()
    pad = torch.zeros(R, dtype=torch.bool)
    pad[torch.randperm(R, generator=g)[:137]] = True
    pad = pad.cuda()
    lab2 = torch.where(pad, torch.full_like(labels, -1), labels)
    out, dl, db = H.box_loss(logits, breg, lab2, regt, n_rows=(lab2 >= 0).sum())
    keep = ~pad
    out0, dl0, db0 = H.box_loss(logits[keep], breg[keep], labels[keep], regt[keep])
    torch.cuda.synchronize()
    assert torch.allclose(out, out0, rtol=1e-6, atol=1e-7)
    assert torch.equal(dl[keep], dl0) and torch.equal(db[keep], db0)
    assert float(dl[pad].abs().max()) == 0.0 and float(db[pad].abs().max()) == 0.0
    # all rows counted: the plain entry point, bit for bit
    o1, d1, b1 = H.box_loss(logits, breg, labels, regt, n_rows=(labels >= 0).sum())
    o2, d2, b2 = H.box_loss(logits, breg, labels, regt)
    assert torch.allclose(o1, o2, rtol=1e-6, atol=0) and torch.equal(d1, d2) and torch.equal(b1, b2)   # (the sums: one atomic per block)
