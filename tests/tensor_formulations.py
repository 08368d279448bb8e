"""The plain tensor formulations (sequences of library tensor calls) of the proposal-pipeline stages that csrc/select.hip and
mmt_box_decode replaced -- kept HERE, as the reference the new kernels are tested bit-for-bit against ("each
new kernel bit-exact against the tensor formulation it replaces").  They restate rpn/inference.py:78-243 and
balanced_positive_negative_sampler.py:20-72 with fixed shapes, as the product ran them before."""
import torch


def decode(codes, boxes, weights, clip):
    w = (boxes[:, 2] - boxes[:, 0] + 1)[:, None]
    h = (boxes[:, 3] - boxes[:, 1] + 1)[:, None]
    cx = boxes[:, 0, None] + 0.5 * w
    cy = boxes[:, 1, None] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = codes[:, 0::4] / wx, codes[:, 1::4] / wy
    dw = torch.clamp(codes[:, 2::4] / ww, max=clip)
    dh = torch.clamp(codes[:, 3::4] / wh, max=clip)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    out = torch.zeros_like(codes)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1
    out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


def rpn_candidates(objectness, regression, anchors, pre_n, clip, lims):
    """per level: top-k of the objectness, gather, decode, clip -> boxes (N, sumk, 4), scores, idx, reg, level offsets.
    objectness[l] (N,A,H,W), regression[l] (N,4A,H,W), anchors[l] (HWA,4); lims (N,2) = (w-1, h-1)"""
    N = objectness[0].shape[0]
    bs, ss, ii, rr, offs = [], [], [], [], [0]
    for o, r, anc in zip(objectness, regression, anchors):
        of = o.permute(0, 2, 3, 1).reshape(N, -1)
        rf = r.permute(0, 2, 3, 1).reshape(N, -1, 4)
        k = min(pre_n, of.shape[1])
        lg, idx = of.topk(k, dim=1, sorted=True)
        rg = torch.gather(rf, 1, idx[:, :, None].expand(-1, -1, 4))
        an = anc[idx.reshape(-1)].view(N, k, 4)
        bx = decode(rg.reshape(-1, 4), an.reshape(-1, 4), (1.0, 1.0, 1.0, 1.0), clip).view(N, k, 4)
        l4 = torch.cat([lims, lims], 1)
        bx = torch.minimum(bx.clamp(min=0), l4[:, None, :])
        bs.append(bx), ss.append(torch.sigmoid(lg)), ii.append(idx), rr.append(rg)
        offs.append(offs[-1] + k)
    return torch.cat(bs, 1), torch.cat(ss, 1), torch.cat(ii, 1), torch.cat(rr, 1), offs


def rpn_post_select(scores, keep, cnt, offs, own_pre, post_n, fpn_post_n, training):
    """-> per image: LongTensor of candidate indices (into that image's sumk candidates), in output order.
    keep (N*L, kmax) int32 positions, cnt (N*L,)"""
    N, sumk = scores.shape
    L = len(offs) - 1
    kmax = keep.shape[1]
    dev = scores.device
    keep = keep.view(N, L, kmax).long()
    cnt = cnt.view(N, L)
    own = torch.tensor(own_pre, device=dev)[None, :, None]
    lvl_off = torch.tensor(offs[:-1], device=dev)[None, :, None]
    ar = torch.arange(kmax, device=dev)[None, None, :]
    valid = (ar < cnt[:, :, None]) & (keep < own)
    if post_n > 0:
        valid = valid & (torch.cumsum(valid.to(torch.int32), 2) <= post_n)
    pos = lvl_off + keep
    kept = torch.zeros((N, sumk + 1), dtype=torch.bool, device=dev)
    kept.scatter_(1, torch.where(valid, pos, torch.full_like(pos, sumk)).view(N, -1), True)
    kept = kept[:, :sumk] & (scores >= 0)
    masked = torch.where(kept, scores, torch.full_like(scores, -1.0))
    out = []
    if training:
        k = min(fpn_post_n, int(kept.sum()))
        top = masked.view(-1).topk(k, sorted=True)[1]
        sel = torch.zeros(N * sumk, dtype=torch.bool, device=dev)
        sel[top] = True
        sel = sel.view(N, sumk) & kept
        for n in range(N):
            out.append(torch.nonzero(sel[n]).squeeze(1))
    else:
        for n in range(N):
            k = min(fpn_post_n, int(kept[n].sum()))
            out.append(masked[n].topk(k, sorted=True)[1])
    return out


def sample_fg_bg(labels, keys, batch, max_pos):
    """one image: -> pos mask, neg mask (bool) -- the num_pos / num_neg members with the smallest keys"""
    pos, neg = labels >= 1, labels == 0
    num_pos = min(int(pos.sum()), max_pos)
    num_neg = min(int(neg.sum()), batch - num_pos)

    def take(member, count):
        out = torch.zeros_like(member)
        if count > 0:
            k = torch.where(member, keys, torch.full_like(keys, 2.0))
            out[torch.topk(k, count, largest=False, sorted=True)[1]] = True
        return out

    return take(pos, num_pos), take(neg, num_neg)


def roi_format_levels(boxes, k_min, k_max, s0=224, lvl0=4, eps=1e-6):
    """Pooler.convert_to_roi_format + LevelMapper (modeling/poolers.py:11-32, 91-104) as the tensor calls mmt_roi_format_levels
    replaced: boxes = per-image (n_i, 4) xyxy tensors -> rois (K, 5), levels (K,) int32"""
    bb = torch.cat(boxes, 0)
    ids = torch.cat([torch.full((len(b), 1), i, dtype=bb.dtype, device=bb.device) for i, b in enumerate(boxes)], 0)
    area = torch.cat([(b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) for b in boxes])
    s = torch.sqrt(area)
    lv = torch.floor(lvl0 + torch.log2(s / s0 + eps))
    lv = torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - k_min
    return torch.cat([ids, bb], 1), lv.to(torch.int32)


# ---- IR-Net (round 4: moved here from the product, which runs csrc/relation.hip only; VERDICT r3 "remove the second backend")
def position_matrix(boxes, dim_g, wave_len):
    """relation_module.py:393-431 (extract_multi_position_matrix, USE_IOU / CLS_WISE_RELATION off): boxes (n, C, 4) ->
    (C, n, n, dim_g) as ~30 elementwise tensor calls -- what `mmt_position_embedding` computes in one launch"""
    import torch as T
    boxes = boxes.permute(1, 0, 2)
    x_min, y_min, x_max, y_max = T.chunk(boxes, 4, dim=2)
    cx, cy = (x_min + x_max) * 0.5, (y_min + y_max) * 0.5
    w, h = (x_max - x_min) + 1., (y_max - y_min) + 1.
    dx = T.log(T.clamp(T.abs((cx - cx.permute(0, 2, 1)) / w), min=1e-3))
    dy = T.log(T.clamp(T.abs((cy - cy.permute(0, 2, 1)) / h), min=1e-3))
    dw = T.log(w / w.permute(0, 2, 1))
    dh = T.log(h / h.permute(0, 2, 1))
    size = dh.size()
    pm = T.stack((dx.view(size), dy.view(size), dw.view(size), dh.view(size)), -1)
    feat_range = T.arange(dim_g / 8, device=boxes.device)
    dim_mat = 1. / (T.pow(wave_len, feat_range / (dim_g / 8)))
    mul = (100. * pm[..., None] * dim_mat.view(1, 1, 1, 1, -1)).view(size[0], size[1], size[2], -1)
    return T.cat((T.sin(mul), T.cos(mul)), -1)


def relation_attention(mod, f_a, position_embedding):
    """RelationModule.forward (reference relation_module.py:33-90) with batched library GEMMs, top-k, softmax and scatter, on the
    parameters of `mod` (the product's RelationModule: its Linear layers run on the HIP GEMM either way)"""
    import math
    import torch.nn.functional as F
    N, ncls, feat_dim = f_a.size()
    g = mod.group
    f_a = f_a.permute(1, 0, 2)
    fr = f_a.contiguous().view(N * ncls, feat_dim)
    w_g = F.relu(mod.WG(position_embedding.reshape(-1, mod.fc_dim[0]).contiguous()))
    w_k = mod.WK(fr).view(-1, N, g, mod.dim_group[1]).permute(0, 2, 3, 1).contiguous().view(-1, mod.dim_group[1], N)
    w_q = mod.WQ(fr).view(-1, N, g, mod.dim_group[0]).transpose(1, 2).contiguous().view(-1, N, mod.dim_group[0])
    aff = (1.0 / math.sqrt(float(mod.dim_group[1]))) * torch.bmm(w_q, w_k)
    w_g = w_g.view(-1, N, N, mod.fc_dim[1]).permute(0, 3, 1, 2).contiguous().view(-1, N, N)
    w_mn = torch.log(torch.clamp(w_g, min=1e-6)) + aff
    k = min(N, mod.topk)
    tv, ti = torch.topk(w_mn, k, dim=2, largest=True, sorted=True)
    w = torch.zeros_like(w_mn).scatter(2, ti, F.softmax(tv, dim=2)).view(ncls, -1, N)
    out = torch.bmm(w, f_a).view(ncls, mod.fc_dim[1], N, feat_dim).permute(1, 3, 2, 0).contiguous()
    # conv1: nn.Conv2d(16 * feat_dim, dim[2], 1, groups=16) as 16 small GEMMs
    x = out.view(1, mod.fc_dim[1] * feat_dim, N, -1)
    _, cin, n, m = x.shape
    o = mod.conv1.weight.shape[0]
    y = torch.bmm(mod.conv1.weight.reshape(g, o // g, cin // g), x.view(g, cin // g, n * m)) + mod.conv1.bias.view(g, o // g, 1)
    return y.view(1, o, n, m).squeeze(0).permute(1, 2, 0)


def ciam(gamma, x, group=None):
    """CIAM_Module.forward (reference mask_relation_module.py:199-242) as two batched library GEMMs; `group`: the attention runs
    inside every group of equal ids (cross-group entries masked to -inf: the same softmax over the same values)"""
    import torch.nn.functional as F
    n, C, Hh, Ww = x.size()
    cw = x.permute(1, 0, 2, 3).reshape(C, n, -1)
    energy = torch.bmm(cw, cw.permute(0, 2, 1))
    if group is not None:
        same = group[:, None] == group[None, :]
        energy = torch.where(same[None], energy, torch.full_like(energy, float("-inf")))
    ne = torch.max(energy, -1, keepdim=True)[0] - energy
    m = torch.mean(ne, 0)
    if group is not None:
        m = torch.where(same, m, torch.full_like(m, float("-inf")))
    att = F.softmax(m, dim=-1)
    out = torch.mm(att, x.reshape(n, -1)).view(n, C, Hh, Ww)
    return gamma * out + x


def _first_argmax(x, dim):
    """numpy.argmax semantics (FIRST maximal index) on the device, whatever the reduction order"""
    mx = x.max(dim=dim, keepdim=True)[0]
    n = x.shape[dim]
    shape = [1] * x.dim()
    shape[dim] = n
    ar = torch.arange(n, device=x.device).view(shape).expand_as(x)
    return torch.where(x == mx, ar, torch.full_like(ar, n)).min(dim=dim)[0].clamp(max=n - 1)


def relation_reg_labels(sorted_boxes, sorted_score, tb, labels, fg_class, target_thresh):
    """DuplicationRemovalNetwork.prepare_reg_label (reference relation_module.py:323-391: a D2H copy and numpy loops per class,
    `eye[argmax]`, `np.intersect1d`) as device tensor arithmetic with numpy's first-index tie rules -- what
    `mmt_relation_reg_labels` computes in one launch per image.  Checked against the oracle's numpy restatement on CPU tensors
    (tests/test_irnet_gpu.py)."""
    import torch.nn.functional as F
    n, G = sorted_boxes.shape[0], tb.shape[0]
    dev = sorted_boxes.device
    if G == 0:
        return torch.zeros((n, fg_class, len(target_thresh)), device=dev)
    a2 = (tb[:, 2] - tb[:, 0] + 1) * (tb[:, 3] - tb[:, 1] + 1)
    ar_g = torch.arange(G, device=dev)
    per_cls = []
    for i in range(fg_class):
        cm = labels == (i + 1)
        score = sorted_score[:, i:i + 1]
        boxes = sorted_boxes[:, i, :]
        a1 = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
        lt = torch.max(boxes[:, None, :2], tb[:, :2])
        rb = torch.min(boxes[:, None, 2:], tb[:, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        iou = inter / (a1[:, None] + a2 - inter)
        iou_c = torch.where(cm[None, :], iou, torch.full_like(iou, -1.0))
        best_gt = F.one_hot(_first_argmax(iou_c, 1), G).to(iou.dtype)
        outs = []
        for th in target_thresh:
            mask = ((iou > th) & cm[None, :]).to(iou.dtype)
            osc = score * mask * best_gt
            oiou = iou * mask * best_gt
            msi = _first_argmax(osc, 0)
            moi = oiou[msi, ar_g]
            valid = mask.sum(1) > 0
            first = torch.full((n + 1,), G, dtype=torch.long, device=dev)
            first = first.scatter_reduce(0, torch.where(cm, msi, torch.full_like(msi, n)), ar_g, reduce="amin",
                                         include_self=True)[:n]
            take = valid & (first < G)
            outs.append(torch.where(take, moi[first.clamp(max=G - 1)], torch.zeros((), device=dev)))
        per_cls.append(torch.stack(outs, -1))
    return torch.stack(per_cls, 1).float()


# ---- box head (round 4: moved here from the product, which runs mmt_det_postprocess only)
def det_filter_results(prob, dec, per, sizes, score_thresh, nms_thresh, detections_per_img, nms_batched):
    """PostProcessor.filter_results (reference box_head/inference.py:87-145) per (image, class): threshold, stable descending
    sort, NMS (`nms_batched`: the product's batched NMS launch), survivors in ascending original-row order, kthvalue cut with
    ties kept.  prob (R, nc), dec (R, nc * 4) decoded + clipped boxes, per = rows per image -> [(boxes, scores, labels)]"""
    dev, nc = prob.device, prob.shape[1]
    segs = []
    for pr, bx in zip(prob.split(per, 0), dec.split(per, 0)):
        for j in range(1, nc):
            sc = pr[:, j]
            masked = torch.where(sc > score_thresh, sc, torch.full_like(sc, -1.0))
            ss, order = torch.sort(masked, descending=True, stable=True)
            segs.append((bx[:, j * 4:(j + 1) * 4][order], ss, order))
    n_valid = torch.stack([(s[1] >= 0).sum() for s in segs]).tolist()
    bl, offs = [], [0]
    for s, nv in zip(segs, n_valid):
        bl.append(s[0][:nv])
        offs.append(offs[-1] + nv)
    kmax = max(max(n_valid), 1)
    keep, cnt = nms_batched(torch.cat(bl, 0), torch.tensor(offs, dtype=torch.int32, device=dev), kmax, nms_thresh)
    cnts = cnt.tolist()
    results, si = [], 0
    for _ in per:
        parts = []
        for j in range(1, nc):
            bxs, ss, order = segs[si]
            kp = keep[si, :cnts[si]].long()
            kp = torch.sort(order[kp])[0]  # `_C.nms` returns ascending ORIGINAL indices (nms_cpu.cpp:64)
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.numel(), device=dev)
            sel = inv[kp]
            parts.append((bxs[sel], ss[sel], torch.full((len(kp),), j, dtype=torch.int64, device=dev)))
            si += 1
        bb = torch.cat([p[0] for p in parts], 0)
        sc = torch.cat([p[1] for p in parts], 0)
        lb = torch.cat([p[2] for p in parts], 0)
        n = bb.shape[0]
        if n > detections_per_img > 0:
            thr = torch.kthvalue(sc, n - detections_per_img + 1)[0]
            k = torch.nonzero(sc >= thr).squeeze(1)
            bb, sc, lb = bb[k], sc[k], lb[k]
        results.append((bb, sc, lb))
    return results
