"""IR-Net relation NMS (reference: modeling/relation/relation_module.py:13-682) -- SURVEY.md row a26.

`DuplicationRemovalNetwork` ranks the top FIRST_N (90) boxes per foreground class, embeds rank + appearance,
runs one multi-head geometric relation attention (`RelationModule`) and regresses each box's IoU with its
ground truth (REG_IOU); at inference those scores replace greedy NMS.  Same parameter names as the reference
(`relation_nms.{nms_rank_fc,roi_feat_embedding_fc,classifier}`, `relation_nms.relation_module.{WG,WK,WQ,conv1}`).

Shipped configuration only (configs/pap/e2e_mask_rcnn_R_50_FPN_1x.yaml + scripts/train_mt.sh): REG_IOU True,
USE_IOU False, CLASS_AGNOSTIC False, CLS_WISE_RELATION False, MERGE_METHOD 0.

Execution: the Linear layers run on the MFMA GEMM (layers.Linear); the attention itself (<= 128 boxes per class: scores,
top-40 softmax, mix, with the 16-group 1x1 conv folded into the value projection) is one launch forward and two backward
(csrc/relation.hip, layers/fused.py::RelationAttentionFn); the numpy label preparation of the reference
(relation_module.py:323-391, a D2H copy + host loops per class) is one launch per image (mmt_relation_reg_labels), the
geometric embedding one launch (mmt_position_embedding).  These kernels are the only implementation in the package: a shape
beyond their capacity (more than 128 ranked boxes per class) or a CPU tensor RAISES; the formulations with library tensor
calls they are tested against live in tests/tensor_formulations.py.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from maskrcnn_benchmark import _hip as _H
from maskrcnn_benchmark.layers import Linear, fused as _fused, nms as _box_nms
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.boxlist_ops import cat_boxlist


_RANK_EMB = {}


def extract_rank_embedding(rank_dim, feat_dim, wave_length=1000, device="cpu"):
    """sin / cos embedding of the ranks 0..rank_dim-1: a constant of (rank_dim, feat_dim, device), computed once"""
    key = (int(rank_dim), int(feat_dim), wave_length, str(device))
    hit = _RANK_EMB.get(key)
    if hit is None:
        hit = _RANK_EMB[key] = _rank_embedding(rank_dim, feat_dim, wave_length, device)
    return hit


def _rank_embedding(rank_dim, feat_dim, wave_length, device):
    rank_range = torch.arange(0, rank_dim, device=device).float()
    feat_range = torch.arange(feat_dim / 2, device=device)
    dim_mat = 1. / (torch.pow(wave_length, feat_range / (feat_dim / 2)))
    mul = rank_range.view(-1, 1) * dim_mat.view(1, -1)
    return torch.cat((torch.sin(mul), torch.cos(mul)), -1)


_POS_FREQ = {}


def extract_multi_position_matrix(boxes, iou, dim_g, wave_len, clswise=False):
    """relation_module.py:393-431: boxes (n, C, 4) -> (C, n, n, dim_g), one launch (mmt_position_embedding)"""
    if iou is not None or clswise:
        raise NotImplementedError("USE_IOU / CLS_WISE_RELATION are off in the shipped configuration")
    if not boxes.is_cuda or boxes.requires_grad or dim_g % 8:
        raise RuntimeError("extract_multi_position_matrix: detached GPU boxes and dim_g % 8 == 0 (the HIP kernel is the only path)")
    from maskrcnn_benchmark import _hip as H
    key = (int(dim_g), wave_len, boxes.device)
    freq = _POS_FREQ.get(key)
    if freq is None:
        feat_range = torch.arange(dim_g / 8, device=boxes.device)
        freq = _POS_FREQ[key] = (1. / (torch.pow(wave_len, feat_range / (dim_g / 8)))).float().contiguous()
    return H.position_embedding(boxes, dim_g, freq)


class _GroupedPointwise(nn.Module):
    """the parameters of nn.Conv2d(in, out, 1, groups=g) of RelationModule.conv1 (names and shapes kept for checkpoints)"""

    def __init__(self, in_ch, out_ch, groups):
        super().__init__()
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch // groups, 1, 1))
        self.bias = nn.Parameter(torch.zeros(out_ch))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    # (no forward: the grouped convolution is folded into the value projection of the attention kernel, see RelationModule)


class RelationModule(nn.Module):
    def __init__(self, appearance_feature_dim=1024, geo_feature_dim=64, fc_dim=(64, 16), group=16,
                 dim=(1024, 1024, 1024), topk=10, iou_method="b"):
        super().__init__()
        assert fc_dim[1] == group, "Check the dimensions in attention!"
        self.fc_dim, self.dim, self.group, self.topk = fc_dim, dim, group, topk
        self.dim_group = (dim[0] // group, dim[1] // group, dim[2] // group)
        self.WG = Linear(geo_feature_dim, fc_dim[1])
        self.WK = Linear(appearance_feature_dim, dim[1])
        self.WQ = Linear(appearance_feature_dim, dim[0])
        self.conv1 = _GroupedPointwise(fc_dim[1] * appearance_feature_dim, dim[2], group)

    def forward(self, f_a, position_embedding, iou=None):
        N, ncls, feat_dim = f_a.size()
        g = self.group
        f_a = f_a.permute(1, 0, 2)
        fr = f_a.contiguous().view(N * ncls, feat_dim)
        dv = self.dim[2] // g
        if not (f_a.is_cuda and self.dim[2] == feat_dim and self.dim[0] == self.dim[1]
                and _H.relation_attention_fits(N, g, self.dim_group[0], dv)):
            raise RuntimeError("RelationModule: GPU tensors, at most %d ranked boxes per class, head width <= 128 / 16 -- the "
                               "attention kernel (csrc/relation.hip) is the only implementation" % _H.RELATION_ATTENTION_MAX_N)
        # one launch each way (csrc/relation.hip).  conv1 (16 groups of feat_dim -> dim[2] / 16) is applied to the
        # features BEFORE the attention mixes them -- sum_m w[n, m] (W_g f[m]) instead of W_g (sum_m w[n, m] f[m]): the
        # same bilinear form -- so it is one plain Linear over the (class, box) rows and the head outputs are dv wide
        w_g = self.WG(position_embedding.reshape(-1, self.fc_dim[0]).contiguous(), relu=True)     # (C N N, 16), (c, n, m) rows
        v = _fused.linear(fr, self.conv1.weight.view(self.dim[2], feat_dim))
        return _fused.RelationAttentionFn.apply(self.WQ(fr), self.WK(fr), w_g, v, self.conv1.bias, ncls, N, g,
                                                min(N, self.topk), 1.0 / math.sqrt(float(self.dim_group[1])))


class DuplicationRemovalNetwork(nn.Module):
    def __init__(self, cfg, is_teacher=False):
        super().__init__()
        self.cfg = cfg.clone()
        r = cfg.MODEL.RELATION_NMS
        if not r.REG_IOU or r.USE_IOU or r.CLASS_AGNOSTIC or r.CLS_WISE_RELATION:
            raise NotImplementedError("relation NMS: only the shipped REG_IOU / per-class configuration is built")
        self.first_n = r.FIRST_N
        self.target_thresh = tuple(r.THREAD)
        self.geo_feature_dim = r.GEO_FEAT_DIM
        self.nms_rank_fc = Linear(r.ROI_FEAT_DIM, r.APPEARANCE_FEAT_DIM)
        self.roi_feat_embedding_fc = Linear(r.ROI_FEAT_DIM, r.APPEARANCE_FEAT_DIM)
        self.relation_module = RelationModule(r.APPEARANCE_FEAT_DIM, geo_feature_dim=self.geo_feature_dim,
                                              fc_dim=(self.geo_feature_dim, 16), group=r.GROUP, dim=tuple(r.HID_DIM),
                                              topk=r.TOPK, iou_method=r.IOU_METHOD)
        self.classifier = Linear(128, len(self.target_thresh))
        self.boxcoder = BoxCoder(weights=(10., 10., 5., 5.))
        self.fg_class = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES - 1
        self.fg_thread = r.FG_THREAD
        self.detections_per_img = cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG
        self.nms = r.POS_NMS
        self.merge_method = r.MERGE_METHOD
        self.roi_feat_dim, self.app_dim = r.ROI_FEAT_DIM, r.APPEARANCE_FEAT_DIM
        self.mode = None

    def set_teacher_mode(self, mode):
        self.mode = mode

    # ---- label preparation on the device (relation_module.py:323-391)
    def prepare_reg_label(self, sorted_boxes, sorted_score, targets):
        """The reference copies to the host and loops over classes with numpy (`eye[argmax]`, `np.intersect1d`); here one launch
        per image (`mmt_relation_reg_labels`: every class against ALL ground-truth boxes of the image with the other classes'
        columns masked out, numpy's first-index tie rules kept), nothing read back.  The tensor formulation it is tested
        against: tests/tensor_formulations.py::relation_reg_labels."""
        labels = targets.get_field("labels")
        tb = targets.bbox
        n, G = sorted_boxes.shape[0], tb.shape[0]
        if G == 0:
            return torch.zeros((n, self.fg_class, len(self.target_thresh)), device=sorted_boxes.device)
        from maskrcnn_benchmark import _hip as H
        out = H.relation_reg_labels(sorted_boxes, sorted_score, tb, labels, self.target_thresh)
        if out is None:
            raise RuntimeError("relation NMS labels: at most 128 ranked boxes, 256 ground-truth boxes and 4 thresholds per image "
                               "(mmt_relation_reg_labels is the only implementation)")
        return out

    def filter_results(self, boxes, targets, scores, image_shape, num_classes, obj):
        """rank the boxes of one image per class (relation_module.py:503-587)"""
        fg = num_classes - 1
        boxes = boxes.reshape(-1, 4 * num_classes)
        bx = torch.stack([boxes[:, j * 4:(j + 1) * 4] for j in range(1, num_classes)], dim=2)  # [R,4,fg]
        sc = scores.reshape(-1, num_classes)[:, 1:]
        first_n = min(bx.shape[0], self.first_n)
        ss, ind = torch.topk(sc, first_n, dim=0, largest=True, sorted=True)
        ori = sc[ind]
        sobj = obj[ind]
        sb = bx[ind]  # [n, fg, 4, fg]
        m = torch.arange(0, fg, device=boxes.device).view(1, -1, 1, 1).expand(first_n, fg, 4, 1)
        sb = torch.gather(sb, 3, m).squeeze(3)
        b = BoxList(sb.reshape(first_n * fg, 4), image_shape, mode="xyxy")
        b.add_field("sorted_idx", ind)
        b.add_field("objectness", sobj.reshape(first_n * fg))
        b.add_field("scores", ss)
        b.add_field("all_scores", ori)
        if self.training:
            b.add_field("labels_iou_reg", self.prepare_reg_label(sb, ss, targets))
        return b.clip_to_image(remove_empty=False)

    def forward(self, x):
        """one image (the reference's call, generalized_rcnn.py:74-85): (fc7 features, [proposals], class probabilities, box
        regression, [target]) -> ([detections], {}) in eval / (None, {"nms_loss"}) in training"""
        appearance_feature, proposals, cls_score, box_reg, targets = x
        assert len(proposals) == 1, "called per image (generalized_rcnn.py:74-85)"
        out, loss = self.forward_batch(appearance_feature, proposals, cls_score, box_reg, targets)
        return (None, loss) if self.training else (out, {})

    def forward_batch(self, appearance_feature, proposals, cls_score, box_reg, targets):
        """All images of the batch at once.  The relation attention works per class on the n = FIRST_N ranked boxes of that
        class, classes never mix -- so the (image, class) pairs of the whole batch are simply more "classes" of one pass:
        one embedding GEMM over all ROIs, one attention over (n, B * fg), one classifier GEMM (the reference loops over
        images, generalized_rcnn.py:74-85).  -> (list of detections per image, {}) / (None, {"nms_loss": mean over images})"""
        fg, T = self.fg_class, len(self.target_thresh)
        sizes = [len(p) for p in proposals]
        tg = targets if targets is not None else [None] * len(sizes)
        with torch.no_grad():
            sbls = []
            for p, cs, br, t in zip(proposals, cls_score.split(sizes), box_reg.split(sizes), tg):
                dec = self.boxcoder.decode(br.detach().reshape(len(p), -1), p.bbox)
                sbls.append(self.filter_results(dec, t, cs.detach(), p.size, fg + 1, p.get_field("objectness")))
        ns = [s.get_field("sorted_idx").shape[0] for s in sbls]
        if len(set(ns)) != 1:  # an image with fewer than FIRST_N proposals: image by image
            outs, losses, st = [], [], 0
            for i, n_i in enumerate(sizes):
                o, l = self._relate(appearance_feature[st:st + n_i], [proposals[i]], [sbls[i]])
                st += n_i
                outs += o if o is not None else []
                losses.append(l)
            if self.training:
                return None, {"nms_loss": torch.mean(torch.stack([l["nms_loss"] for l in losses]))}
            return outs, {}
        return self._relate(appearance_feature, proposals, sbls)

    def _relate(self, appearance_feature, proposals, sbls):
        fg, T, B = self.fg_class, len(self.target_thresh), len(sbls)
        n = sbls[0].get_field("sorted_idx").shape[0]
        emb = self.roi_feat_embedding_fc(appearance_feature, input_relu=True, in_mask_scale=getattr(self, "in_mask_scale", 1.0))
        app, st = [], 0
        for p, s_ in zip(proposals, sbls):
            app.append(emb[st + s_.get_field("sorted_idx")])        # (n, fg, app_dim)
            st += len(p)
        app = torch.cat(app, 1) if B > 1 else app[0]                # (n, B * fg, app_dim)
        rank = self.nms_rank_fc(extract_rank_embedding(n, self.roi_feat_dim, device=app.device))
        sf = app + rank[:, None, :]
        bb = [s_.bbox.reshape(-1, fg, 4) for s_ in sbls]
        pos = extract_multi_position_matrix(torch.cat(bb, 1) if B > 1 else bb[0], None, self.geo_feature_dim, 1000)
        sf = F.relu(sf + self.relation_module(sf, pos, None))
        sf_all = self.classifier(sf.reshape(-1, self.app_dim).contiguous()).view(n, B * fg, T)
        if self.training:
            losses = [F.mse_loss(s_.get_field("labels_iou_reg"), sf_all[:, i * fg:(i + 1) * fg]) for i, s_ in enumerate(sbls)]
            return None, {"nms_loss": torch.mean(torch.stack(losses)) if B > 1 else losses[0]}
        return [self._detect(p, s_, sf_all[:, i * fg:(i + 1) * fg]) for i, (p, s_) in enumerate(zip(proposals, sbls))], {}

    def _detect(self, p, sbl, sf):
        """inference tail of one image (relation_module.py:250-321): regressed scores -> per-class NMS -> detections"""
        fg = self.fg_class
        scores = sbl.get_field("scores")
        bboxes = sbl.bbox.reshape(-1, fg, 4)
        n = scores.shape[0]
        sc3 = torch.cat([scores[:, :, None]] * len(self.target_thresh), dim=-1)
        with torch.no_grad():
            s = (sf * (sc3 > self.fg_thread).float())[:, :, min(max(self.merge_method, 0), len(self.target_thresh) - 1)]
            objectness = sbl.get_field("objectness").reshape(-1, fg)
            all_scores = sbl.get_field("all_scores")
            # nuclei (class index 1, label 2, NMS 0.5) first, then cytoplasm (class 0, label 1, POS_NMS) (:261-312).  Fixed
            # shapes: every class keeps its n ranked boxes; the ones below FG_THREAD are moved far away with score -1 so they
            # neither suppress nor survive; both classes go through ONE batched NMS (sorted by the regressed score, as
            # `_C.nms` sorts internally); survivors are emitted in rank order (`_C.nms` returns ascending indices).
            order_cls = ((1, 2, 0.5), (0, 1, self.nms))
            if any(not t for _, _, t in order_cls):
                raise NotImplementedError("relation NMS: a disabled per-class NMS is not on the shipped recipe")
            valid = torch.stack([s[:, c] >= self.fg_thread for c, _, _ in order_cls])            # [2, n]
            cs = torch.stack([s[:, c] for c, _, _ in order_cls])
            bx = torch.stack([bboxes[:, c, :] for c, _, _ in order_cls])                           # [2, n, 4]
            key = torch.where(valid, cs, torch.full_like(cs, -1.0))
            so = torch.sort(key, dim=1, descending=True, stable=True)[1]                           # [2, n]
            bs = torch.gather(bx, 1, so[:, :, None].expand(-1, -1, 4))
            vs = torch.gather(valid, 1, so)
            bs = torch.where(vs[:, :, None], bs, torch.full_like(bs, -1.0e6))
            from maskrcnn_benchmark import _hip as H
            from maskrcnn_benchmark.utils.miscellaneous import dev_const
            seg1 = dev_const([0, n], torch.int32, bs.device)
            kc = [H.nms_batched(bs[q].contiguous(), seg1, n, thr) for q, (_, _, thr) in enumerate(order_cls)]  # thresholds differ
            keep, cnt = torch.cat([k for k, _ in kc], 0), torch.cat([c for _, c in kc], 0)
            kept_sorted = torch.zeros((2, n + 1), dtype=torch.bool, device=bs.device)
            kj = torch.where(torch.arange(n, device=bs.device)[None, :] < cnt[:, None], keep.long(),
                             torch.full_like(keep, n, dtype=torch.int64))
            kept_sorted.scatter_(1, kj, True)
            kept = torch.zeros((2, n), dtype=torch.bool, device=bs.device)
            kept.scatter_(1, so, kept_sorted[:, :n] & vs)                                          # back to rank order
            # the only read-back of the module: how many detections each class keeps (<= 2 n = 180 <= DETECTIONS_PER_IMG)
            if 2 * n > self.detections_per_img > 0:
                raise NotImplementedError("relation NMS: FIRST_N * classes > DETECTIONS_PER_IMG needs the kthvalue cut")
            counts = kept.sum(1).tolist()
            parts = []
            for q, (cls, lab, _) in enumerate(order_cls):
                index = torch.nonzero_static(kept[q], size=counts[q]).squeeze(1)
                b = BoxList(bboxes[index, cls, :], p.size, mode="xyxy")
                b.add_field("scores", s[index, cls])
                b.add_field("objectness", objectness[index, cls])
                b.add_field("all_scores", all_scores[index, cls])
                b.add_field("labels", torch.full((counts[q],), lab, dtype=torch.int64, device=s.device))
                parts.append(b)
            r = cat_boxlist(parts)
        return r
