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
