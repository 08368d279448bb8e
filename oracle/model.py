"""ORACLE -- test infrastructure only (see oracle/__init__.py).

Pure-torch, CPU, NCHW, functional restatement of the reference's Mask R-CNN R50-FPN
mean-teacher hot path.  Operates on a plain state-dict (reference key names)
and a plain config object; consumes the torch global RNG in the same order as the reference so
that, with the same seed, sampled index sets and dropout masks coincide.

Every function cites the reference lines it restates (paths relative to
/root/reference/maskrcnn_benchmark/).  `taps` (a dict) records the discrete decisions
(proposals, sampled indices, dropout masks, detections) so GPU parity tests can replay them.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import native


# --------------------------------------------------------------------------------------------
# effective config (configs/pap/e2e_mask_rcnn_R_50_FPN_1x.yaml over config/defaults.py,
# MT flags of scripts/train_mt.sh:4-20; IR-Net off)
# --------------------------------------------------------------------------------------------
def default_cfg(**over):
    c = dict(
        anchor_sizes=(32, 64, 128, 256, 512), anchor_strides=(4, 8, 16, 32, 64),
        aspect_ratios=(0.5, 1.0, 2.0), straddle_thresh=0,
        rpn_fg_iou=0.7, rpn_bg_iou=0.3, rpn_batch=256, rpn_pos_frac=0.5,
        pre_nms_train=2000, pre_nms_test=1000, post_nms_train=2000, post_nms_test=1000,
        fpn_post_nms_train=2000, fpn_post_nms_test=1000, rpn_nms_thresh=0.7, rpn_min_size=0,
        roi_fg_iou=0.5, roi_bg_iou=0.5, bbox_reg_weights=(10.0, 10.0, 5.0, 5.0),
        roi_batch=512, roi_pos_frac=0.25, score_thresh=0.05, det_nms=0.5, dets_per_img=200,
        box_res=7, mask_res=14, pool_scales=(0.25, 0.125, 0.0625, 0.03125), pool_sr=2,
        num_classes=3, mask_out=28, dropout=0.5, size_div=32,
        mt_temp=0.5, mt_sharpen=True, mt_hard_neg=True, mt_cls_balance=1.5, mt_rank_filter=0.2,
        mt_cls_loss_type="bce", mt_cls_loss=0.2, mt_fg_hint=1.0, mt_lambda=5.0, mt_start=1000,
        mt_alpha=0.99, mt_rampup=250, mt_rampdown=250, nms_loss_w=1.0, mask_thresh=0.5,
        relation=False,  # IR-Net (RELATION_NMS + RELATION_MASK), oracle/irnet.py
    )
    c.update(over)
    return SimpleNamespace(**c)


class Boxes(object):
    """Minimal BoxList (structures/bounding_box.py:9-266): xyxy float32 + fields + (W,H)."""

    def __init__(self, bbox, size, fields=None):
        self.bbox = bbox
        self.size = (int(size[0]), int(size[1]))
        self.fields = dict(fields or {})

    def __len__(self):
        return self.bbox.shape[0]

    def index(self, idx):
        out = Boxes(self.bbox[idx], self.size)
        for k, v in self.fields.items():
            if isinstance(v, list):
                ii = idx.nonzero().squeeze(1).tolist() if idx.dtype in (torch.bool, torch.uint8) else idx.tolist()
                out.fields[k] = [v[i] for i in ii]
            else:
                out.fields[k] = v[idx]
        return out

    def area(self):  # bounding_box.py:240-250
        b = self.bbox
        return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)

    def clip(self):  # bounding_box.py:229-238 (remove_empty=False; in place)
        w, h = self.size
        self.bbox[:, 0].clamp_(min=0, max=w - 1)
        self.bbox[:, 1].clamp_(min=0, max=h - 1)
        self.bbox[:, 2].clamp_(min=0, max=w - 1)
        self.bbox[:, 3].clamp_(min=0, max=h - 1)
        return self

    def hflip(self):  # bounding_box.py:133-169 FLIP_LEFT_RIGHT
        w = self.size[0]
        x1, y1, x2, y2 = self.bbox.split(1, dim=-1)
        out = Boxes(torch.cat((w - x2 - 1, y1, w - x1 - 1, y2), dim=-1), self.size)
        out.fields = dict(self.fields)
        return out


def cat_boxes(lst):  # structures/boxlist_ops.py:106-134
    out = Boxes(torch.cat([b.bbox for b in lst], 0), lst[0].size)
    for k in lst[0].fields:
        if k == "mask":
            continue
        out.fields[k] = torch.cat([b.fields[k] for b in lst], 0)
    return out


def box_iou(a, b):  # structures/boxlist_ops.py:57-92
    area1, area2 = a.area(), b.area()
    lt = torch.max(a.bbox[:, None, :2], b.bbox[:, :2])
    rb = torch.min(a.bbox[:, None, 2:], b.bbox[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


# ------------------------------------------------------------------------------ box coder
BBOX_XFORM_CLIP = math.log(1000.0 / 16)


def box_encode(ref, prop, weights):  # modeling/box_coder.py:22-50
    ew = prop[:, 2] - prop[:, 0] + 1
    eh = prop[:, 3] - prop[:, 1] + 1
    ex = prop[:, 0] + 0.5 * ew
    ey = prop[:, 1] + 0.5 * eh
    gw = ref[:, 2] - ref[:, 0] + 1
    gh = ref[:, 3] - ref[:, 1] + 1
    gx = ref[:, 0] + 0.5 * gw
    gy = ref[:, 1] + 0.5 * gh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (gx - ex) / ew, wy * (gy - ey) / eh,
                        ww * torch.log(gw / ew), wh * torch.log(gh / eh)), dim=1)


def box_decode(codes, boxes, weights):  # modeling/box_coder.py:52-95
    boxes = boxes.to(codes.dtype)
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx = codes[:, 0::4] / wx
    dy = codes[:, 1::4] / wy
    dw = torch.clamp(codes[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(codes[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * w[:, None] + cx[:, None]
    pcy = dy * h[:, None] + cy[:, None]
    pw = torch.exp(dw) * w[:, None]
    ph = torch.exp(dh) * h[:, None]
    out = torch.zeros_like(codes)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw - 1
    out[:, 3::4] = pcy + 0.5 * ph - 1
    return out


# ------------------------------------------------------------------------------ anchors
def cell_anchors(stride, size, ratios):  # modeling/rpn/anchor_generator.py:196-265
    scales = np.array([size], dtype=np.float64) / stride
    ratios = np.array(ratios, dtype=np.float64)
    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1

    def whctr(a):
        w = a[2] - a[0] + 1
        h = a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws = ws[:, None]
        hs = hs[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    w, h, xc, yc = whctr(base)
    sr = (w * h) / ratios
    ws = np.round(np.sqrt(sr))
    hs = np.round(ws * ratios)
    ra = mk(ws, hs, xc, yc)
    out = []
    for i in range(ra.shape[0]):
        w, h, xc, yc = whctr(ra[i])
        out.append(mk(w * scales, h * scales, xc, yc))
    return torch.from_numpy(np.vstack(out)).float()


def make_anchors(cfg, image_sizes, feat_shapes):
    """-> per image: list over levels of Boxes with field 'visibility'
    (anchor_generator.py:65-123).  image_sizes are (H, W)."""
    per_level = []
    for (gh, gw), stride, size in zip(feat_shapes, cfg.anchor_strides, cfg.anchor_sizes):
        base = cell_anchors(stride, size, cfg.aspect_ratios)
        sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx = xx.reshape(-1)
        yy = yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        per_level.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    out = []
    for (ih, iw) in image_sizes:
        lv = []
        for a in per_level:
            t = cfg.straddle_thresh
            vis = (a[:, 0] >= -t) & (a[:, 1] >= -t) & (a[:, 2] < iw + t) & (a[:, 3] < ih + t)
            lv.append(Boxes(a, (iw, ih), {"visibility": vis}))
        out.append(lv)
    return out


# ------------------------------------------------------------------------------ matcher / sampler
def matcher(mq, hi, lo, allow_low):  # modeling/matcher.py:44-139 (top_k < 2 branch)
    if mq.numel() == 0:
        raise ValueError("No ground-truth / proposal boxes available for one of the images during training")
    vals, matches = mq.max(dim=0)
    allm = matches.clone()
    below = vals < lo
    between = (vals >= lo) & (vals < hi)
    matches[below] = -1
    matches[between] = -2
    if allow_low:
        best, _ = mq.max(dim=1)
        pairs = torch.nonzero(mq == best[:, None])
        upd = pairs[:, 1]
        matches[upd] = allm[upd]
    return matches


def fg_bg_sampler(labels_list, batch, frac, taps, tag):
    """modeling/balanced_positive_negative_sampler.py:20-72.  Two torch.randperm draws per image
    from the global CPU generator, in this order."""
    pos_out, neg_out = [], []
    for li, lab in enumerate(labels_list):
        positive = torch.nonzero(lab >= 1).squeeze(1)
        negative = torch.nonzero(lab == 0).squeeze(1)
        num_pos = min(positive.numel(), int(batch * frac))
        num_neg = min(negative.numel(), batch - num_pos)
        perm1 = torch.randperm(positive.numel())[:num_pos]
        perm2 = torch.randperm(negative.numel())[:num_neg]
        pi, ni = positive[perm1], negative[perm2]
        pm = torch.zeros_like(lab, dtype=torch.bool)
        nm = torch.zeros_like(lab, dtype=torch.bool)
        pm[pi] = True
        nm[ni] = True
        pos_out.append(pm)
        neg_out.append(nm)
        if taps is not None:
            taps.setdefault(tag, []).append((pm.nonzero().squeeze(1).clone(), nm.nonzero().squeeze(1).clone()))
    return pos_out, neg_out


def smooth_l1(inp, tgt, beta, size_average=True):  # layers/smooth_l1_loss.py:6-20
    n = torch.abs(inp - tgt)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()


# ------------------------------------------------------------------------------ backbone
def _fbn(sd, p):  # layers/batch_norm.py:19-24 (no eps)
    scale = sd[p + ".weight"] * sd[p + ".running_var"].rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return scale.reshape(1, -1, 1, 1), bias.reshape(1, -1, 1, 1)


def _conv_bn(sd, x, conv, bn, stride=1, pad=0):
    y = F.conv2d(x, sd[conv + ".weight"], None, stride, pad)
    s, b = _fbn(sd, bn)
    return y * s + b


BLOCKS = (3, 4, 6, 3)


def resnet_body(sd, x, pre="backbone.body."):  # backbone/resnet.py:117-124, 254-274, 288-293
    x = F.relu(_conv_bn(sd, x, pre + "stem.conv1", pre + "stem.bn1", 2, 3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li, nb in enumerate(BLOCKS, 1):
        for b in range(nb):
            p = "%slayer%d.%d." % (pre, li, b)
            stride = 2 if (b == 0 and li > 1) else 1
            o = F.relu(_conv_bn(sd, x, p + "conv1", p + "bn1", stride, 0))  # STRIDE_IN_1X1
            o = F.relu(_conv_bn(sd, o, p + "conv2", p + "bn2", 1, 1))
            o = _conv_bn(sd, o, p + "conv3", p + "bn3", 1, 0)
            res = x
            if (p + "downsample.0.weight") in sd:
                res = _conv_bn(sd, x, p + "downsample.0", p + "downsample.1", stride, 0)
            x = F.relu(o + res)
        outs.append(x)
    return outs


def fpn(sd, cs, pre="backbone.fpn."):  # backbone/fpn.py:43-74
    def conv(name, x, pad):
        return F.conv2d(x, sd[pre + name + ".weight"], sd[pre + name + ".bias"], 1, pad)

    last = conv("fpn_inner4", cs[3], 0)
    res = [conv("fpn_layer4", last, 1)]
    for i in (3, 2, 1):
        td = F.interpolate(last, scale_factor=2, mode="nearest")
        lat = conv("fpn_inner%d" % i, cs[i - 1], 0)
        last = lat + td
        res.insert(0, conv("fpn_layer%d" % i, last, 1))
    res.append(F.max_pool2d(res[-1], 1, 2, 0))
    return tuple(res)


def backbone(sd, x):
    return fpn(sd, resnet_body(sd, x))


def rpn_head(sd, feats, pre="rpn.head."):  # rpn/rpn.py:39-46
    logits, regs = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, sd[pre + "conv.weight"], sd[pre + "conv.bias"], 1, 1))
        logits.append(F.conv2d(t, sd[pre + "cls_logits.weight"], sd[pre + "cls_logits.bias"]))
        regs.append(F.conv2d(t, sd[pre + "bbox_pred.weight"], sd[pre + "bbox_pred.bias"]))
    return logits, regs


def hint_adaptor(sd, feats, pre="hint_adaptor."):  # box_head/roi_box_feature_extractors.py:45-75
    return [F.conv2d(f, sd["%sadapter_%d.weight" % (pre, i + 1)], sd["%sadapter_%d.bias" % (pre, i + 1)])
            for i, f in enumerate(feats)]


def to_image_list(imgs, div):  # structures/image_list.py:39-80
    """imgs: list of (3,H,W) tensors or a (N,3,H,W) tensor -> (padded tensor, [(H,W)])"""
    if isinstance(imgs, torch.Tensor):
        imgs = list(imgs)
    mh = max(i.shape[1] for i in imgs)
    mw = max(i.shape[2] for i in imgs)
    if div > 0:
        mh = int(math.ceil(mh / div) * div)
        mw = int(math.ceil(mw / div) * div)
    out = imgs[0].new_zeros((len(imgs), 3, mh, mw))
    for i, im in enumerate(imgs):
        out[i, :, :im.shape[1], :im.shape[2]].copy_(im)
    return out, [(int(i.shape[1]), int(i.shape[2])) for i in imgs]


# ------------------------------------------------------------------------------ RPN post-processing
def _flatten_level(obj, reg):
    N, A, H, W = obj.shape
    o = obj.permute(0, 2, 3, 1).reshape(N, -1)
    r = reg.view(N, -1, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4)
    return o, r


def rpn_postprocess(cfg, anchors, objectness, regression, train_cfg, module_training, targets=None,
                    is_teacher=False):
    """rpn/inference.py:78-172,216-243.  train_cfg selects the *_TRAIN/_TEST numbers (which
    selector object), module_training is nn.Module.training of the selector."""
    pre_n = cfg.pre_nms_train if train_cfg else cfg.pre_nms_test
    post_n = cfg.post_nms_train if train_cfg else cfg.post_nms_test
    fpn_post = cfg.fpn_post_nms_train if train_cfg else cfg.fpn_post_nms_test
    n_img = len(anchors)
    per_level = []
    for lvl, (obj, reg) in enumerate(zip(objectness, regression)):
        N = obj.shape[0]
        o, r = _flatten_level(obj, reg)
        o = o.sigmoid()
        k = min(pre_n, o.shape[1])
        o, idx = o.topk(k, dim=1, sorted=True)
        bi = torch.arange(N)[:, None]
        r = r[bi, idx]
        anc = torch.cat([anchors[i][lvl].bbox for i in range(n_img)], 0).reshape(N, -1, 4)[bi, idx]
        props = box_decode(r.view(-1, 4), anc.view(-1, 4), (1.0, 1.0, 1.0, 1.0)).view(N, -1, 4)
        res = []
        for i in range(N):
            b = Boxes(props[i], anchors[i][lvl].size, {"objectness": o[i]})
            if is_teacher:
                b.fields["box_reg"] = r[i]
                b.fields["rpn_topk"] = idx[i]
                b.fields["rpn_ancher_level"] = torch.tensor([lvl] * idx.shape[1])
            b.clip()
            ws = b.bbox[:, 2] - b.bbox[:, 0] + 1
            hs = b.bbox[:, 3] - b.bbox[:, 1] + 1
            b = b.index(((ws >= cfg.rpn_min_size) & (hs >= cfg.rpn_min_size)).nonzero().squeeze(1))
            keep = native.nms(b.bbox, b.fields["objectness"], cfg.rpn_nms_thresh)
            if post_n > 0:
                keep = keep[:post_n]
            res.append(b.index(keep))
        per_level.append(res)
    boxlists = [cat_boxes([per_level[l][i] for l in range(len(per_level))]) for i in range(n_img)]
    if len(objectness) > 1:
        if module_training:
            allo = torch.cat([b.fields["objectness"] for b in boxlists], 0)
            sizes = [len(b) for b in boxlists]
            k = min(fpn_post, len(allo))
            _, inds = torch.topk(allo, k, dim=0, sorted=True)
            m = torch.zeros_like(allo, dtype=torch.bool)
            m[inds] = True
            ms = m.split(sizes)
            boxlists = [boxlists[i].index(ms[i]) for i in range(n_img)]
        else:
            for i in range(n_img):
                o = boxlists[i].fields["objectness"]
                k = min(fpn_post, len(o))
                _, inds = torch.topk(o, k, dim=0, sorted=True)
                boxlists[i] = boxlists[i].index(inds)
    if module_training and targets is not None:  # add_gt_proposals :55-76
        out = []
        for b, t in zip(boxlists, targets):
            gt = Boxes(t.bbox, t.size, {"objectness": torch.ones(len(t))})
            out.append(cat_boxes([b, gt]))
        boxlists = out
    return boxlists


def rpn_targets(cfg, anchors, targets):  # rpn/loss.py:42-83
    labels, regs = [], []
    for a, t in zip(anchors, targets):
        m = matcher(box_iou(t, a), cfg.rpn_fg_iou, cfg.rpn_bg_iou, True)
        mt = t.bbox[m.clamp(min=0)]
        lab = (m >= 0).to(torch.float32)
        lab[~a.fields["visibility"]] = -1
        lab[m == -2] = -1
        labels.append(lab)
        regs.append(box_encode(mt, a.bbox, (1.0, 1.0, 1.0, 1.0)))
    return labels, regs


def rpn_loss(cfg, anchors, objectness, regression, targets, taps=None):  # rpn/loss.py:138-196
    anc = [cat_boxes(a) for a in anchors]
    labels, regt = rpn_targets(cfg, anc, targets)
    pos, neg = fg_bg_sampler(labels, cfg.rpn_batch, cfg.rpn_pos_frac, taps, "rpn_sampler")
    pos = torch.nonzero(torch.cat(pos, 0)).squeeze(1)
    neg = torch.nonzero(torch.cat(neg, 0)).squeeze(1)
    samp = torch.cat([pos, neg], 0)
    of, rf = [], []
    for o, r in zip(objectness, regression):
        o2, r2 = _flatten_level(o, r)
        of.append(o2)
        rf.append(r2)
    obj = torch.cat(of, 1).reshape(-1)
    reg = torch.cat(rf, 1).reshape(-1, 4)
    labels = torch.cat(labels, 0)
    regt = torch.cat(regt, 0)
    box_loss = smooth_l1(reg[pos], regt[pos], 1.0 / 9, size_average=False) / samp.numel()
    obj_loss = F.binary_cross_entropy_with_logits(obj[samp], labels[samp])
    return obj_loss, box_loss


# ------------------------------------------------------------------------------ poolers / heads
def level_map(boxes_list, k_min, k_max):  # modeling/poolers.py:31-42
    s = torch.sqrt(torch.cat([b.area() for b in boxes_list]))
    lv = torch.floor(4 + torch.log2(s / 224 + 1e-6))
    return torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - k_min


def pooler(cfg, feats, boxes_list, res):  # modeling/poolers.py:78-121
    bb = torch.cat([b.bbox for b in boxes_list], 0)
    ids = torch.cat([torch.full((len(b), 1), i, dtype=bb.dtype) for i, b in enumerate(boxes_list)], 0)
    rois = torch.cat([ids, bb], 1)
    k_min = -math.log2(cfg.pool_scales[0])
    k_max = -math.log2(cfg.pool_scales[-1])
    lv = level_map(boxes_list, k_min, k_max)
    out = torch.zeros((len(rois), feats[0].shape[1], res, res), dtype=feats[0].dtype)
    for l, sc in enumerate(cfg.pool_scales):
        idx = torch.nonzero(lv == l).squeeze(1)
        out[idx] = native.roi_align(feats[l], rois[idx], (res, res), sc, cfg.pool_sr)
    return out


def box_feature(sd, cfg, feats, boxes_list, istrain, taps=None, pre="box_heads.box.feature_extractor."):
    """box_head/roi_box_feature_extractors.py:107-125"""
    x = pooler(cfg, feats, boxes_list, cfg.box_res)
    x = x.view(x.size(0), -1)
    x = F.relu(F.linear(x, sd[pre + "fc6.weight"], sd[pre + "fc6.bias"]))
    x = F.relu(F.linear(x, sd[pre + "fc7.weight"], sd[pre + "fc7.bias"]))
    if cfg.dropout > 0 and istrain:
        # F.dropout(training=True) == x * bernoulli(1-p) / (1-p); draw explicitly so the mask is tappable
        mask = torch.empty_like(x).bernoulli_(1 - cfg.dropout)
        if taps is not None:
            taps.setdefault("dropout", []).append(mask.clone())
        x = x * mask / (1 - cfg.dropout)
    return x


def box_predictor(sd, x, pre="box_heads.box.predictor."):  # roi_box_predictors.py:47-51
    return (F.linear(x, sd[pre + "cls_score.weight"], sd[pre + "cls_score.bias"]),
            F.linear(x, sd[pre + "bbox_pred.weight"], sd[pre + "bbox_pred.bias"]))


def box_subsample(cfg, proposals, targets, taps=None):  # box_head/loss.py:42-116
    labels, regs = [], []
    for p, t in zip(proposals, targets):
        m = matcher(box_iou(t, p), cfg.roi_fg_iou, cfg.roi_bg_iou, False)
        mi = m.clamp(min=0)
        lab = t.fields["labels"][mi].to(torch.int64)
        lab[m == -1] = 0
        lab[m == -2] = -1
        labels.append(lab)
        regs.append(box_encode(t.bbox[mi], p.bbox, cfg.bbox_reg_weights))
    pos, neg = fg_bg_sampler(labels, cfg.roi_batch, cfg.roi_pos_frac, taps, "roi_sampler")
    out = []
    for p, lab, rg, pm, nm in zip(proposals, labels, regs, pos, neg):
        q = Boxes(p.bbox, p.size, p.fields)
        q.fields["labels"] = lab
        q.fields["regression_targets"] = rg
        out.append(q.index(torch.nonzero(pm | nm).squeeze(1)))
    return out


def fastrcnn_loss(class_logits, box_regression, proposals):  # box_head/loss.py:118-162
    labels = torch.cat([p.fields["labels"] for p in proposals], 0)
    regt = torch.cat([p.fields["regression_targets"] for p in proposals], 0)
    cls = F.cross_entropy(class_logits, labels)
    pos = torch.nonzero(labels > 0).squeeze(1)
    lp = labels[pos]
    mi = 4 * lp[:, None] + torch.tensor([0, 1, 2, 3])
    bl = smooth_l1(box_regression[pos[:, None], mi], regt[pos], 1, size_average=False)
    return cls, bl / labels.numel()


def box_postprocess(cfg, class_logits, box_regression, proposals):  # box_head/inference.py:36-145
    prob = F.softmax(class_logits, -1)
    per = [len(p) for p in proposals]
    cat = torch.cat([p.bbox for p in proposals], 0)
    dec = box_decode(box_regression.view(sum(per), -1), cat, cfg.bbox_reg_weights)
    nc = prob.shape[1]
    out = []
    for pr, bx, p in zip(prob.split(per, 0), dec.split(per, 0), proposals):
        bl = Boxes(bx.reshape(-1, 4), p.size, {"scores": pr.reshape(-1)})
        bl.clip()
        boxes = bl.bbox.reshape(-1, nc * 4)
        scores = bl.fields["scores"].reshape(-1, nc)
        res = []
        for j in range(1, nc):
            inds = (scores[:, j] > cfg.score_thresh).nonzero().squeeze(1)
            sj = scores[inds, j]
            bj = boxes[inds, j * 4:(j + 1) * 4]
            keep = native.nms(bj, sj, cfg.det_nms)
            res.append(Boxes(bj[keep], p.size, {"scores": sj[keep], "objectness": sj[keep],
                                                "labels": torch.full((len(keep),), j, dtype=torch.int64)}))
        r = cat_boxes(res)
        n = len(r)
        if n > cfg.dets_per_img > 0:
            thr, _ = torch.kthvalue(r.fields["scores"], n - cfg.dets_per_img + 1)
            r = r.index(torch.nonzero(r.fields["scores"] >= thr.item()).squeeze(1))
        out.append(r)
    return out


def mask_feature(sd, cfg, feats, boxes_list, pre="mask_heads.mask.feature_extractor."):
    """mask_head/roi_mask_feature_extractors.py:131-146"""
    x = pooler(cfg, feats, boxes_list, cfg.mask_res)
    for i in range(1, 5):
        x = F.relu(F.conv2d(x, sd["%smask_fcn%d.weight" % (pre, i)], sd["%smask_fcn%d.bias" % (pre, i)], 1, 1))
    return x


def mask_predictor(sd, x, pre="mask_heads.mask.predictor."):  # roi_mask_predictors.py:34-36
    x = F.relu(F.conv_transpose2d(x, sd[pre + "conv5_mask.weight"], sd[pre + "conv5_mask.bias"], 2, 0))
    return F.conv2d(x, sd[pre + "mask_fcn_logits.weight"], sd[pre + "mask_fcn_logits.bias"])


def project_masks_on_boxes(polys, boxes, M):
    """mask_head/loss.py:37-75 + structures/segmentation_mask.py:96-133.
    polys: list (per positive) of list of 1-D float32 tensors; boxes float32 (P,4)."""
    out = []
    for inst, box in zip(polys, boxes):
        w, h = box[2] - box[0], box[3] - box[1]
        w = max(w, 1)
        h = max(h, 1)
        cropped = []
        for p in inst:
            q = p.clone()
            q[0::2] = q[0::2] - box[0]
            q[1::2] = q[1::2] - box[1]
            cropped.append(q)
        rw, rh = float(M) / float(w), float(M) / float(h)
        scaled = []
        for q in cropped:
            if rw == rh:
                scaled.append(q * rw)
            else:
                q = q.clone()
                q[0::2] *= rw
                q[1::2] *= rh
                scaled.append(q)
        out.append(torch.from_numpy(native.poly_mask([s.numpy() for s in scaled], M, M)))
    if not out:
        return torch.empty(0, dtype=torch.float32)
    return torch.stack(out, 0).to(torch.float32)


def mask_loss(cfg, proposals, mask_logits, targets, taps=None):  # mask_head/loss.py:119-180
    labels, mts = [], []
    for p, t in zip(proposals, targets):
        m = matcher(box_iou(t, p), cfg.roi_fg_iou, cfg.roi_bg_iou, False)
        mi = m.clamp(min=0)
        lab = t.fields["labels"][mi].to(torch.int64)
        lab[m == -1] = 0
        pos = torch.nonzero(lab > 0).squeeze(1)
        polys = [t.fields["masks"][i] for i in mi[pos].tolist()]
        mts.append(project_masks_on_boxes(polys, p.bbox[pos], cfg.mask_out))
        labels.append(lab)
    labels = torch.cat(labels, 0)
    mt = torch.cat(mts, 0)
    if taps is not None:
        taps["mask_targets"] = mt.clone()
    pos = torch.nonzero(labels > 0).squeeze(1)
    if mt.numel() == 0:
        return mask_logits.sum() * 0
    return F.binary_cross_entropy_with_logits(mask_logits[pos, labels[pos]], mt)


def paste_mask(mask, box, im_h, im_w, thresh=0.5, padding=1):  # mask_head/inference.py:169-206
    M = mask.shape[-1]
    pad2 = 2 * padding
    scale = float(M + pad2) / M
    pm = mask.new_zeros((1, 1, M + pad2, M + pad2))
    pm[:, :, padding:-padding, padding:-padding] = mask
    wh = (box[2] - box[0]) * .5
    hh = (box[3] - box[1]) * .5
    xc = (box[2] + box[0]) * .5
    yc = (box[3] + box[1]) * .5
    wh = wh * scale
    hh = hh * scale
    b = torch.stack([xc - wh, yc - hh, xc + wh, yc + hh]).to(torch.int32)
    w = max(int(b[2] - b[0] + 1), 1)
    h = max(int(b[3] - b[1] + 1), 1)
    m = F.interpolate(pm.to(torch.float32), size=(h, w), mode="bilinear", align_corners=False)[0][0]
    m = m > thresh
    im = torch.zeros((im_h, im_w), dtype=torch.uint8)
    x0 = max(int(b[0]), 0)
    x1 = min(int(b[2]) + 1, im_w)
    y0 = max(int(b[1]), 0)
    y1 = min(int(b[3]) + 1, im_h)
    im[y0:y1, x0:x1] = m[(y0 - int(b[1])):(y1 - int(b[1])), (x0 - int(b[0])):(x1 - int(b[0]))]
    return im


def mask_generate(cfg, mask_logits, dets):
    """MaskPostProcessor + Masker in teacher 'test' mode (mask_head/inference.py:29-65,209-246)
    -> per image integral mask (H,W) int64 == t.get_field('mask').sum(0)[0]
    (detector/generalized_rcnn.py:129-132)."""
    prob = mask_logits.sigmoid()
    labels = torch.cat([d.fields["labels"] for d in dets])
    prob = prob[torch.arange(prob.shape[0]), labels][:, None]
    out = []
    for pr, d in zip(prob.split([len(d) for d in dets], 0), dets):
        iw, ih = d.size
        res = [paste_mask(m[0], b, ih, iw, cfg.mask_thresh, 1) for m, b in zip(pr, d.bbox)]
        if not res:
            raise ValueError("no detections for an unlabeled image (reference: bare except, step skipped)")
        out.append(torch.stack(res, 0)[:, None].sum(0)[0])
    return out


# ------------------------------------------------------------------------------ the three forwards
def forward_supervised(sd, cfg, images, targets, taps=None):
    """GeneralizedRCNN.forward in training mode (detector/generalized_rcnn.py:42-115), IR-Net off.
    images (N,3,H,W) or list; targets list[Boxes] with fields labels (int64), masks (polygons)."""
    x, sizes = to_image_list(images, cfg.size_div) if not isinstance(images, tuple) else images
    feats = backbone(sd, x)
    obj, reg = rpn_head(sd, feats)
    anchors = make_anchors(cfg, sizes, [f.shape[-2:] for f in feats])
    with torch.no_grad():
        props = rpn_postprocess(cfg, anchors, [o.detach() for o in obj], [r.detach() for r in reg],
                                True, True, targets)
    if taps is not None:
        taps["features"] = [f.detach() for f in feats]
        taps["rpn_objectness"] = [o.detach() for o in obj]
        taps["rpn_regression"] = [r.detach() for r in reg]
        taps["rpn_proposals"] = [(p.bbox.clone(), p.fields["objectness"].clone()) for p in props]
    l_obj, l_rpn = rpn_loss(cfg, anchors, obj, reg, targets, taps)
    with torch.no_grad():
        samp = box_subsample(cfg, props, targets, taps)
    xf = box_feature(sd, cfg, feats, samp, True, taps)
    cl, br = box_predictor(sd, xf)
    l_cls, l_box = fastrcnn_loss(cl, br, samp)
    extra = {}
    if cfg.relation:  # generalized_rcnn.py:63-95
        from . import irnet
        per = [len(s_) for s_ in samp]
        probs = F.softmax(cl, dim=1)
        nl = [irnet.dup_removal(sd, cfg, xi, si, ci, bi, ti, True)[1]
              for xi, si, ci, bi, ti in zip(xf.split(per), samp, probs.split(per), br.split(per), targets)]
        extra["nms_loss"] = torch.mean(torch.stack(nl))
    pos = [s.index(torch.nonzero(s.fields["labels"] > 0).squeeze(1)) for s in samp]
    mx = mask_feature(sd, cfg, feats, pos)
    ml = mask_predictor(sd, mx)
    l_seg = mask_loss(cfg, pos, ml, targets, taps)
    if cfg.relation:  # mask_head.py:96-147 (DEEP_SUPER)
        from . import irnet
        perp = [len(p) for p in pos]
        outs = [irnet.mask_relation(sd, cfg, f, m, p) for f, m, p in zip(mx.split(perp), ml.split(perp), pos)]
        ml2 = torch.cat([o[0] for o in outs])
        l_seg = 0.5 * (l_seg + mask_loss(cfg, [o[1] for o in outs], ml2, targets))
    if taps is not None:
        taps["class_logits"] = cl.detach()
        taps["box_regression"] = br.detach()
        taps["mask_logits"] = ml.detach()
        taps["sampled"] = [(s.bbox.clone(), s.fields["labels"].clone(), s.fields["regression_targets"].clone())
                           for s in samp]
    out = {"loss_classifier": l_cls, "loss_box_reg": l_box, "loss_seg": l_seg,
           "loss_objectness": l_obj, "loss_rpn_box_reg": l_rpn}
    out.update(extra)
    return out


def inference(sd, cfg, x, sizes, taps=None):
    """GeneralizedRCNN.forward in eval mode with module mode 'test' (teacher coarse inference,
    generalized_rcnn.py:122-132): -> detections list[Boxes], integral masks."""
    feats = backbone(sd, x)
    obj, reg = rpn_head(sd, feats)
    anchors = make_anchors(cfg, sizes, [f.shape[-2:] for f in feats])
    props = rpn_postprocess(cfg, anchors, obj, reg, False, False)
    xf = box_feature(sd, cfg, feats, props, False)
    cl, br = box_predictor(sd, xf)
    if cfg.relation:
        from . import irnet
        per = [len(p) for p in props]
        probs = F.softmax(cl, dim=1)
        dets = [irnet.dup_removal(sd, cfg, xi, pi, ci, bi, None, False)[0]
                for xi, pi, ci, bi in zip(xf.split(per), props, probs.split(per), br.split(per))]
    else:
        dets = box_postprocess(cfg, cl, br, props)
    dets_in = dets  # what the mask head receives (before the per-class objectness sort of the mask relation)
    mx = mask_feature(sd, cfg, feats, dets)
    ml = mask_predictor(sd, mx)
    if cfg.relation:
        perd = [len(d) for d in dets]
        outs = [irnet.mask_relation(sd, cfg, f, m, d) for f, m, d in zip(mx.split(perd), ml.split(perd), dets)]
        ml = torch.cat([o[0] for o in outs])
        dets = [o[1] for o in outs]
    seg = mask_generate(cfg, ml, dets)
    if taps is not None:
        taps["infer_proposals"] = [(p.bbox.clone(), p.fields["objectness"].clone()) for p in props]
        taps["detections"] = [(d.bbox.clone(), d.fields["scores"].clone(), d.fields["labels"].clone(),
                               d.fields.get("objectness", d.fields["scores"]).clone()) for d in dets_in]
        taps["detections_out"] = [(d.bbox.clone(), d.fields["labels"].clone()) for d in dets]
        taps["infer_mask_logits"] = ml.detach().clone()
    return dets, seg


def forward_teacher(sd, cfg, images, taps=None):
    """GeneralizedRCNN.forward_teacher (generalized_rcnn.py:117-167), targets=None, no_grad,
    model.eval().  images: list of AUG_K inputs (each (N,3,H,W) tensor or list of (3,H,W))."""
    with torch.no_grad():
        ils = [to_image_list(im, cfg.size_div) for im in images]
        x0, sizes = ils[0]
        dets, seg = inference(sd, cfg, x0, sizes, taps)
        feats = []
        for (x, _) in ils:  # extract_aug_feat :201-208
            feats.append(backbone(sd, x))
            feats.append(backbone(sd, torch.flip(x, (3,))))
        f0 = feats[0]
        obj, reg = rpn_head(sd, f0)
        anchors = make_anchors(cfg, sizes, [f.shape[-2:] for f in f0])
        # box_selector_train of an eval() module: TRAIN numbers, test-style per-image top-k, no GT
        props = rpn_postprocess(cfg, anchors, obj, reg, True, False, dets, is_teacher=True)
        # teacher_sample_selection (rpn/loss.py:85-136): result unused, but it draws from the RNG
        anc = [cat_boxes(a) for a in anchors]
        labels, _ = rpn_targets(cfg, anc, dets)
        fg_bg_sampler(labels, cfg.rpn_batch, cfg.rpn_pos_frac, taps, "teacher_rpn_sampler")
        emb = [hint_adaptor(sd, f) for f in feats]
        if taps is not None:
            taps["teacher_proposals"] = [(p.bbox.clone(), p.fields["objectness"].clone()) for p in props]
        samp = box_subsample(cfg, props, dets, taps)
        sampB = [s.hflip() for s in samp]
        logits = []
        for i, f in enumerate(feats):
            xf = box_feature(sd, cfg, f, samp if i % 2 == 0 else sampB, False)
            logits.append(box_predictor(sd, xf)[0])
    return {"result_t": samp, "class_logit_t": logits, "embedding": emb, "seg_mask": seg, "ffi_boxes": None}


def fg_hint_loss(teachers, students, masks):  # detector/generalized_rcnn.py:243-282 (MGD)
    new_t = [[torch.flip(f, (3,)) for f in feat] if i % 2 == 1 else feat for i, feat in enumerate(teachers)]
    sizes = [f.shape for f in students[0]]
    masks = torch.stack(masks)
    mlist = []
    for s in sizes:
        m = F.adaptive_avg_pool2d(masks[:, None, :, :].float(), s[2:])
        m = (m > 0.5).to(m.dtype)
        mlist.append(m)
    ori = students[0::2] if len(students) > 1 else students
    flp = students[1::2] if len(students) > 1 else []
    dists = []
    for feat in new_t:
        for st in ori:
            for sf, tf, mk in zip(st, feat, mlist):
                dists.append((((sf - tf) ** 2) * mk).sum() / (mk.sum() * sf.shape[1] + 1e-7))
    for feat in new_t:
        for st in flp:
            for sf, tf, mk in zip(st, feat, mlist):
                sf = torch.flip(sf, (3,))
                dists.append((((sf - tf) ** 2) * mk).sum() / (mk.sum() * sf.shape[1] + 1e-7))
    return torch.mean(torch.stack(dists))


def sharpen(p, temp):  # box_head/loss.py:311-315
    pt = p ** (1 / temp)
    return (pt / pt.sum(dim=1, keepdim=True)).detach()


def psm_loss(cfg, class_logits, class_logits_t, labels):
    """FastRCNNLossComputation.evaluatePSM (box_head/loss.py:164-237,267-287)."""
    t_logits = torch.mean(torch.stack(class_logits_t), dim=0)
    lg = class_logits_t
    if cfg.mt_cls_loss_type == "bce":
        lg = [F.softmax(l, dim=1) for l in lg]
    m_logit = torch.mean(torch.stack(lg), dim=0)  # _mean_var_logits :164-173: of the PROBABILITIES for 'bce'
    v = torch.std(torch.stack(lg), dim=0)
    pos = torch.nonzero(labels > 0).squeeze(1)
    neg = torch.nonzero(labels == 0).squeeze(1)
    vp = v[pos].sum(-1)
    vn = v[neg].sum(-1)
    tp, tn = t_logits[pos], t_logits[neg]
    losses = []
    for cl in class_logits:
        if cfg.mt_rank_filter > 0:
            if cfg.mt_hard_neg:
                order = torch.argsort(vn, descending=True)
            else:
                order = torch.randperm(vn.shape[0])
            keep = order[:min(order.shape[0], int(vp.shape[0] / 2))]
            tl = torch.cat([tp, tn[keep]])
            sl = torch.cat([cl[pos], cl[neg][keep]])
            pn = (tp.shape[0], keep.shape[0])
        else:
            tl, sl, pn = m_logit, cl, None  # as written (:229): not the mean logits
        w = cfg.mt_cls_balance if cfg.mt_hard_neg else 1
        losses.append(psm_cls_loss(cfg, sl, tl.clone(), pn, w))
    return torch.mean(torch.stack(losses), dim=0)


def psm_cls_loss(cfg, logit, teacher, pn, bal):  # box_head/loss.py:267-287
    typ = {"bce": "ce", "wbce": "wce"}.get(cfg.mt_cls_loss_type, cfg.mt_cls_loss_type)
    if typ == "kl":
        return F.kl_div(F.log_softmax(logit, dim=1), F.softmax(teacher, dim=1), reduction="mean")
    if typ == "mse":
        return F.mse_loss(logit, teacher.detach())
    lp = F.log_softmax(logit, dim=1)
    t = F.softmax(teacher, dim=1)
    if cfg.mt_sharpen:
        t = sharpen(t, cfg.mt_temp)
    if pn is None:
        return (-t.detach() * lp).mean(0).sum() / 3
    w = torch.ones(logit.shape[0])
    w[pn[0]:] = bal
    return (-t.detach() * lp * w[:, None]).mean(0).sum() / 3


def forward_student(sd, cfg, images, tr, taps=None):
    """GeneralizedRCNN.forward_student (generalized_rcnn.py:170-199); images: list of AUG_S inputs."""
    ils = [to_image_list(im, cfg.size_div) for im in images]
    feats = []
    for i, (x, _) in enumerate(ils):
        if i % 2 == 1:
            x = torch.flip(x, (3,))
        feats.append(backbone(sd, x))
    out = {}
    semb = [hint_adaptor(sd, f) for f in feats]
    out["mt_fg_loss"] = fg_hint_loss(tr["embedding"], semb, tr["seg_mask"])
    props = tr["result_t"]
    propsB = [p.hflip() for p in props]
    logits = []
    for i, f in enumerate(feats):
        xf = box_feature(sd, cfg, f, props if i % 2 == 0 else propsB, True, taps)
        logits.append(box_predictor(sd, xf)[0])
    labels = torch.cat([p.fields["labels"] for p in props], 0)
    out["mt_classifier"] = psm_loss(cfg, logits, tr["class_logit_t"], labels)
    if taps is not None:
        taps["student_logits"] = [l.detach() for l in logits]
    return out


# ------------------------------------------------------------------------------ engine arithmetic
def sigmoid_rampup(cur, length):  # utils/miscellaneous.py:233-240
    if length == 0:
        return 1.0
    cur = np.clip(cur, 0.0, length)
    ph = 1.0 - cur / length
    return float(np.exp(-5.0 * ph * ph))


def sigmoid_rampdown(gap, length):  # utils/miscellaneous.py:242-247
    if length == 0:
        return 1.0
    ph = 1.0 - gap / length
    return float(np.exp(-12 * ph * ph))


def mt_weight(cfg, step, total):
    """the `weight` of engine/MTtrainer.py:89-95 (note D10: rampdown is fed the ramp-UP length)"""
    if (step - cfg.mt_start) < cfg.mt_rampup and (step - cfg.mt_start) > 0:
        return cfg.mt_lambda * sigmoid_rampup(step - cfg.mt_start, cfg.mt_rampup)
    if (total - step) < cfg.mt_rampdown:
        return cfg.mt_lambda * sigmoid_rampdown(total - step, cfg.mt_rampup)
    return cfg.mt_lambda


def weight_sum_losses(cfg, loss_dict, step, total):  # engine/MTtrainer.py:67-109
    w = mt_weight(cfg, step, total)
    bal = {"mt_classifier": cfg.mt_cls_loss, "nms_loss": cfg.nms_loss_w, "mt_fg_loss": cfg.mt_fg_hint}
    out = {}
    for k, v in loss_dict.items():
        v = w * v if "mt" in k else v
        out[k] = v * bal[k] if k in bal else v
    return out


def ema_alpha(cfg, it):  # engine/MTtrainer.py:277-278
    return min(1 - 1 / (it + 1), cfg.mt_alpha)


def ema_update(teacher_params, student_params, alpha):  # engine/MTtrainer.py:279-281
    for t, s in zip(teacher_params, student_params):
        t.mul_(alpha).add_(s, alpha=1 - alpha)


# ------------------------------------------------------------------------------ one whole iteration [A]-[E]
def lr_factor(last_epoch, steps=(5000,), gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500, method="linear"):
    """WarmupMultiStepLR.get_lr / base_lr (solver/lr_scheduler.py:40-53)"""
    from bisect import bisect_right
    w = 1
    if last_epoch < warmup_iters:
        if method == "constant":
            w = warmup_factor
        else:
            a = last_epoch / warmup_iters
            w = warmup_factor * (1 - a) + a
    return w * gamma ** bisect_right(list(steps), last_epoch)


class Trainer(object):
    """The body of MTtrainer.train (engine/MTtrainer.py:165-229) on plain state dicts, with the reference's own optimiser
    construction (solver/build.py:5-23: one torch.optim.SGD group per trainable tensor, bias lr x BIAS_LR_FACTOR and
    WEIGHT_DECAY_BIAS) and schedule (scheduler.step() BEFORE optimizer.step(), MTtrainer.py:182).  Gradients come from
    torch autograd through the restated forwards; ROIAlign backward is the gradcheck'd restatement in oracle/native.

    `trainable` / `param_order`: the reference's named_parameters() order and requires_grad set (tests/golden/
    state_shapes.json, captured from the reference model)."""

    def __init__(self, sd, cfg, trainable, param_order, base_lr=0.005, momentum=0.9, weight_decay=1e-4,
                 bias_lr_factor=2, weight_decay_bias=0, max_iter=7000):
        self.cfg, self.max_iter = cfg, max_iter
        self.s = {k: v.clone() for k, v in sd.items()}
        self.t = {k: v.clone() for k, v in sd.items()}
        self.trainable = [k for k in trainable if k in self.s]
        self.param_order = [k for k in param_order if k in self.s]
        groups = []
        for k in self.trainable:
            self.s[k].requires_grad_(True)
            if "box_heads.box.D" in k:  # solver/build.py:11
                continue
            lr, wd = base_lr, weight_decay
            if "bias" in k:
                lr, wd = base_lr * bias_lr_factor, weight_decay_bias
            groups.append({"params": [self.s[k]], "lr": lr, "weight_decay": wd, "initial_lr": lr})
        self.opt = torch.optim.SGD(groups, base_lr, momentum=momentum)
        self.last_epoch = 0  # _LRScheduler.__init__ has stepped once

    def step(self, iteration, images, targets, unlabeled=None, seeds=(None, None, None)):
        """-> (weighted loss dict, taps of the supervised / teacher / student forwards).  `seeds`: torch.manual_seed
        before each of the three forwards (None = leave the global generator alone)."""
        cfg = self.cfg
        taps_a, taps_b, taps_c = {}, {}, {}
        if seeds[0] is not None:
            torch.manual_seed(seeds[0])
        loss = forward_supervised(self.s, cfg, images, targets, taps_a)          # [A] MTtrainer.py:176
        if iteration > cfg.mt_start and cfg.mt_lambda > 0 and unlabeled is not None:  # :177
            if seeds[1] is not None:
                torch.manual_seed(seeds[1])
            k = len(unlabeled) - 1  # AUG_K teacher views, AUG_S = 1 student view
            tr = forward_teacher(self.t, cfg, unlabeled[:k], taps_b)             # [B] :258-261
            if seeds[2] is not None:
                torch.manual_seed(seeds[2])
            loss.update(forward_student(self.s, cfg, unlabeled[-1:], tr, taps_c))  # [C] :266-267
        self.last_epoch += 1                                                     # scheduler.step() :182
        f = lr_factor(self.last_epoch)
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * f
        wl = weight_sum_losses(cfg, loss, iteration, self.max_iter)             # :183
        self.opt.zero_grad()                                                     # [D] :191-193
        sum(wl.values()).backward()
        self.opt.step()
        if cfg.mt_lambda > 0 and iteration > (cfg.mt_start - 10):                # [E] :195-196
            alpha = ema_alpha(cfg, iteration - (cfg.mt_start - 10))
            with torch.no_grad():
                ema_update([self.t[k] for k in self.param_order], [self.s[k].detach() for k in self.param_order], alpha)
        return {k: v.detach() for k, v in wl.items()}, (taps_a, taps_b, taps_c)
