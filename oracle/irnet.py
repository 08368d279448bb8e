"""ORACLE -- test infrastructure only (see oracle/__init__.py).

CPU restatement of IR-Net (BASELINE config 5):
  * DuplicationRemovalNetwork / RelationModule  -- modeling/relation/relation_module.py:13-601
  * extract_rank_embedding / extract_multi_position_matrix -- relation_module.py:604-682
  * MaskRelationRefineNet / CIAM_Module -- modeling/relation/mask_relation_module.py:16-242
  * RoiAlignMaskFeatureExtractor -- modeling/relation/relation_mask_feature_extractor.py:10-48
for the shipped configuration (configs/pap/e2e_mask_rcnn_R_50_FPN_1x.yaml + scripts/train_mt.sh): REG_IOU True,
USE_IOU False, CLASS_AGNOSTIC False, FIRST_N 90, TOPK 40, THREAD (0.1,), FG_THREAD 0.1, POS_NMS 0.55, MERGE_METHOD 0;
mask relation TYPE 'CIAM', NORM -1, PRE_NORM False, EXTRACTOR_CHANNEL 16, SAME_PREDICTOR False, DEEP_SUPER True.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native
from .model import Boxes, box_decode, cat_boxes


def irnet_cfg(cfg):
    d = dict(first_n=90, rel_topk=40, thresholds=(0.1,), fg_thread=0.1, pos_nms=0.55, merge_method=0, geo_dim=64,
             app_dim=128, roi_feat_dim=1024, group=16, hid=(1024, 1024, 128), ciam_topk=128)
    for k, v in d.items():
        if not hasattr(cfg, k):
            setattr(cfg, k, v)
    return cfg


def rank_embedding(rank_dim, feat_dim, wave_length=1000):  # relation_module.py:604-624
    rank_range = torch.arange(0, rank_dim).float()
    feat_range = torch.arange(feat_dim / 2)
    dim_mat = feat_range / (feat_dim / 2)
    dim_mat = 1. / (torch.pow(wave_length, dim_mat))
    mul = rank_range.view(-1, 1) * dim_mat.view(1, -1)
    return torch.cat((torch.sin(mul), torch.cos(mul)), -1)


def position_matrix(boxes, dim_g, wave_len):  # relation_module.py:626-682 (iou None, clswise False)
    boxes = boxes.permute(1, 0, 2)  # [cls, n, 4]
    x_min, y_min, x_max, y_max = torch.chunk(boxes, 4, dim=2)
    cx = (x_min + x_max) * 0.5
    cy = (y_min + y_max) * 0.5
    w = (x_max - x_min) + 1.
    h = (y_max - y_min) + 1.
    dx = torch.log(torch.clamp(torch.abs((cx - cx.permute(0, 2, 1)) / w), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy - cy.permute(0, 2, 1)) / h), min=1e-3))
    dw = torch.log(w / w.permute(0, 2, 1))
    dh = torch.log(h / h.permute(0, 2, 1))
    size = dh.size()
    pm = torch.cat([t.view(size[0], size[1], size[2], 1) for t in (dx, dy, dw, dh)], -1)
    feat_range = torch.arange(dim_g / 8)
    dim_mat = 1. / (torch.pow(wave_len, feat_range / (dim_g / 8)))
    pm = 100. * pm.view(size[0], size[1], size[2], 4, -1)
    mul = (pm * dim_mat.view(1, 1, 1, 1, -1)).view(size[0], size[1], size[2], -1)
    return torch.cat((torch.sin(mul), torch.cos(mul)), -1)


def relation_module(sd, pre, f_a, pos, cfg):  # relation_module.py:33-90
    N, ncls, feat_dim = f_a.size()
    g, dg = cfg.group, (cfg.hid[0] // cfg.group, cfg.hid[1] // cfg.group)
    f_a = f_a.permute(1, 0, 2)
    fr = f_a.contiguous().view(N * ncls, feat_dim)
    w_g = F.relu(F.linear(pos.view(-1, cfg.geo_dim), sd[pre + "WG.weight"], sd[pre + "WG.bias"]))
    w_k = F.linear(fr, sd[pre + "WK.weight"], sd[pre + "WK.bias"]).view(-1, N, g, dg[1]).permute(0, 2, 3, 1)
    w_k = w_k.contiguous().view(-1, dg[1], N)
    w_q = F.linear(fr, sd[pre + "WQ.weight"], sd[pre + "WQ.bias"]).view(-1, N, g, dg[0]).transpose(1, 2)
    w_q = w_q.contiguous().view(-1, N, dg[0])
    aff = (1.0 / math.sqrt(float(dg[1]))) * torch.bmm(w_q, w_k)
    w_g = w_g.view(-1, N, N, g).permute(0, 3, 1, 2).contiguous().view(-1, N, N)
    w_mn = torch.log(torch.clamp(w_g, min=1e-6)) + aff
    top_k = min(N, cfg.rel_topk)
    tv, ti = torch.topk(w_mn, top_k, dim=2, largest=True, sorted=True)
    w = torch.zeros_like(w_mn).scatter(2, ti, F.softmax(tv, dim=2)).view(ncls, -1, N)
    out = torch.bmm(w, f_a).view(ncls, g, N, feat_dim).permute(1, 3, 2, 0).contiguous().view(1, g * feat_dim, N, -1)
    out = F.conv2d(out, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], groups=g)
    return out.squeeze(0).permute(1, 2, 0)  # (N, ncls, hid[2]); squeeze(0) == squeeze() unless N or ncls is 1


def prepare_reg_label(sorted_boxes, sorted_score, target, thresholds):  # relation_module.py:323-391 (numpy as is)
    labels = target.fields["labels"]
    n = sorted_boxes.shape[0]
    out = []
    for i in range(sorted_boxes.shape[1]):
        idx = torch.nonzero(labels == (i + 1))[:, 0]
        tb = target.bbox[idx]
        G = len(idx)
        if G == 0:
            out.append(np.zeros((n, len(thresholds))))
            continue
        eye = np.eye(G)
        score = sorted_score[:, i:i + 1].cpu().numpy()
        boxes = sorted_boxes[:, i, :].reshape(-1, 4)
        a1 = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
        a2 = (tb[:, 2] - tb[:, 0] + 1) * (tb[:, 3] - tb[:, 1] + 1)
        lt = torch.max(boxes[:, None, :2], tb[:, :2])
        rb = torch.min(boxes[:, None, 2:], tb[:, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        iou = (inter / (a1[:, None] + a2 - inter)).cpu().numpy()
        per = []
        for th in thresholds:
            mask = iou > th
            oiou = iou * mask
            valid = np.where(mask)[0]
            osc = np.tile(score, (1, G))
            osc *= mask
            mm = eye[np.argmax(iou, axis=1)]
            osc *= mm
            oiou = oiou * mm
            msi = np.argmax(osc, axis=0)
            moi = oiou[msi, np.arange(osc.shape[1])]
            reg = np.zeros((n,))
            oidx, i1, _ = np.intersect1d(msi, valid, return_indices=True)
            reg[oidx] = moi[i1]
            per.append(reg)
        out.append(np.stack(per, axis=-1))
    return np.stack(out, axis=1).astype(np.float32, copy=False)


def _rank(cfg, boxes, target, scores, size, obj, training):
    """filter_results (relation_module.py:503-587), CLASS_AGNOSTIC False, REG_IOU True"""
    nc = scores.shape[1]
    boxes = boxes.reshape(-1, 4 * nc)
    bx = torch.cat([boxes[:, j * 4:(j + 1) * 4][:, :, None] for j in range(1, nc)], dim=2)  # [R,4,fg]
    sc = scores[:, 1:]
    first_n = min(bx.shape[0], cfg.first_n)
    ss, ind = torch.topk(sc, first_n, dim=0, largest=True, sorted=True)
    ori = sc[ind]        # [n, fg, fg]
    sobj = obj[ind]      # [n, fg]
    sb = bx[ind]         # [n, fg, 4, fg]
    fg = nc - 1
    m = torch.arange(0, fg).view(1, -1, 1, 1).expand(first_n, fg, 4, 1)
    sb = torch.gather(sb, 3, m).squeeze(3)  # [n, fg, 4]
    reg = torch.from_numpy(prepare_reg_label(sb, ss, target, cfg.thresholds)) if training else None
    b = Boxes(sb.reshape(first_n * fg, 4), size, {"sorted_idx": ind, "objectness": sobj.reshape(first_n * fg),
                                                  "scores": ss, "all_scores": ori})
    if reg is not None:
        b.fields["labels_iou_reg"] = reg
    b.clip()
    return b


def dup_removal(sd, cfg, feat, proposal, cls_score, box_reg, target, training, pre="relation_nms."):
    """DuplicationRemovalNetwork.forward for ONE image (it is called per image, generalized_rcnn.py:74-85)
    -> (detections Boxes or None, nms_loss or None)"""
    irnet_cfg(cfg)
    fg = cls_score.shape[1] - 1
    with torch.no_grad():
        dec = box_decode(box_reg.view(len(proposal), -1), proposal.bbox, (10., 10., 5., 5.))
        sbl = _rank(cfg, dec, target, cls_score.detach(), proposal.size, proposal.fields["objectness"], training)
    ind = sbl.fields["sorted_idx"]
    scores = sbl.fields["scores"]
    bboxes = sbl.bbox.reshape(-1, fg, 4)
    objectness = sbl.fields["objectness"].reshape(-1, fg)
    all_scores = sbl.fields["all_scores"]
    n = ind.shape[0]
    app = F.linear(feat, sd[pre + "roi_feat_embedding_fc.weight"], sd[pre + "roi_feat_embedding_fc.bias"])[ind]
    rank = F.linear(rank_embedding(n, cfg.roi_feat_dim), sd[pre + "nms_rank_fc.weight"], sd[pre + "nms_rank_fc.bias"])
    sf = app + rank[:, None, :]
    pos = position_matrix(bboxes, cfg.geo_dim, 1000)
    sf = F.relu(sf + relation_module(sd, pre + "relation_module.", sf, pos, cfg))
    sf = F.linear(sf.view(-1, cfg.app_dim), sd[pre + "classifier.weight"], sd[pre + "classifier.bias"])
    sf = sf.view(-1, fg, len(cfg.thresholds))
    sc3 = torch.cat([scores[:, :, None]] * len(cfg.thresholds), dim=-1)
    if training:
        return None, F.mse_loss(sbl.fields["labels_iou_reg"].float(), sf.float())
    # merge_multi_thread_score_test with MERGE_METHOD 0: thread 0 (relation_module.py:589-601)
    s = (sf * (sc3 > cfg.fg_thread).float())[:, :, 0]
    res = []
    for cls, lab, thr in ((1, 2, 0.5), (0, 1, cfg.pos_nms)):  # nuclei first, then cytoplasm (relation_module.py:261-312)
        index = (s[:, cls] >= cfg.fg_thread).nonzero()[:, 0]
        cs, cb = s[index, cls], bboxes[index, cls, :]
        keep = native.nms(cb, cs, thr) if thr else torch.arange(len(cs))
        res.append(Boxes(cb[keep], proposal.size, {"scores": cs[keep], "objectness": objectness[index, cls][keep],
                                                   "all_scores": all_scores[index, cls][keep],
                                                   "labels": torch.full((len(keep),), lab, dtype=torch.int64)}))
    r = cat_boxes(res)
    nd = len(r)
    if nd > cfg.dets_per_img > 0:
        thr, _ = torch.kthvalue(r.fields["scores"], nd - cfg.dets_per_img + 1)
        r = r.index(torch.nonzero(r.fields["scores"] >= thr.item()).squeeze(1))
    return r, None


def ciam(gamma, x, topk):  # mask_relation_module.py:199-242 (NORM -1, PRE_NORM False)
    n, C, Hh, Ww = x.size()
    cw = x.permute(1, 0, 2, 3)
    q = cw.contiguous().view(C, n, -1)
    k = cw.contiguous().view(C, n, -1).permute(0, 2, 1)
    energy = torch.bmm(q, k)
    ne = torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy
    att = F.softmax(torch.mean(ne, 0), dim=-1)
    out = torch.bmm(att[None, :, :], x.view(1, n, -1)).view(n, C, Hh, Ww)
    return gamma * out + x


def mask_relation(sd, cfg, feat_roi, mask_logits, proposal, pre="mask_heads.mask.mask_relation_module."):
    """MaskRelationRefineNet.forward for ONE image (mask_relation_module.py:53-155), CIAM branch.
    proposal: Boxes with labels / objectness -> (logits_2 in class-sorted order, sorted proposal Boxes)"""
    irnet_cfg(cfg)
    labels, obj = proposal.fields["labels"], proposal.fields["objectness"]
    sel, srt_roi, srt_mask, order = [], [], [], []
    for c in range(cfg.num_classes - 1):
        idx = torch.nonzero(labels == (c + 1))[:, 0]
        _, si = torch.sort(obj[idx], descending=True)
        idx = idx[si]
        sel.append(mask_logits[idx, c + 1])
        srt_mask.append(mask_logits[idx])
        srt_roi.append(feat_roi[idx])
        order.append(idx)
    order = torch.cat(order)
    sorted_mask = torch.cat(srt_mask, 0)
    cls_len = [s.shape[0] for s in sel]
    selm = torch.sigmoid(torch.cat(sel, 0))
    roi = torch.cat(srt_roi, 0)
    fe = pre + "appearance_feature_extractor."
    x = torch.cat((roi, F.max_pool2d(selm[:, None, :, :], kernel_size=2, stride=2)), 1)
    for nm in ("mask_fcn1", "mask_fcn2", "mask_fcn3", "conv5_mask"):
        x = F.relu(F.conv2d(x, sd[fe + nm + ".weight"], sd[fe + nm + ".bias"], 1, 1))
    rel = [ciam(sd[pre + "relation_module.gamma"], f, cfg.ciam_topk) for f in torch.split(x, cls_len) if f.shape[0] != 0]
    rel = torch.cat(rel)
    rel = F.relu(F.conv_transpose2d(rel, sd[pre + "deconv_1.weight"], sd[pre + "deconv_1.bias"], 2, 0))
    rel = F.conv2d(rel, sd[pre + "classifier.weight"], sd[pre + "classifier.bias"])
    sorted_mask = sorted_mask.clone()
    sorted_mask[torch.arange(rel.shape[0])] = rel
    return sorted_mask, proposal.index(order)
