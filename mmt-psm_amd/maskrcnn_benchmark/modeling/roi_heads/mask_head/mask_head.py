"""Mask head (reference: modeling/roi_heads/mask_head/{mask_head,roi_mask_feature_extractors,
roi_mask_predictors,loss,inference}.py).

Module tree kept: feature_extractor.mask_fcn{1-4}, predictor.conv5_mask (2x2 s2 deconv) / mask_fcn_logits.
The two per-ROI CPU Python loops of the reference are single launches here:
  * mask targets  (project_masks_on_boxes + pycocotools rasteriser)  -> `mmt_polygon_targets`
  * teacher pseudo-mask (Masker.paste_mask_in_image per detection)     -> `mmt_paste_masks`
"""
import torch
from torch import nn

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import Conv2d, ConvTranspose2d, fused
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.poolers import Pooler
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
from maskrcnn_benchmark.utils.miscellaneous import dev_ints
from maskrcnn_benchmark.modeling.relation.mask_relation_module import MaskRelationRefineNet


def keep_only_positive_boxes(boxes):
    pos_boxes, pos_inds = [], []
    for b in boxes:
        m = b.get_field("labels") > 0
        n_pos = getattr(b, "n_pos", None)  # set by the box head's sampler: no blocking nonzero here
        na = getattr(b, "n_pos_async", None)
        if n_pos is None and na is not None:
            # round 6 (SURVEY f-2): the count left the device through a pinned buffer when the sampler ran; everything the box head
            # does with the sampled lists has been queued in between -- the wait here is for work long done
            na[1].synchronize()
            n_pos = int(na[0][na[2]])
        pos_boxes.append(b[torch.nonzero_static(m, size=n_pos).squeeze(1) if n_pos is not None else m.nonzero().squeeze(1)])
        pos_inds.append(m)
    return pos_boxes, pos_inds


class MaskRCNNFPNFeatureExtractor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        m = cfg.MODEL.ROI_MASK_HEAD
        self.pooler = Pooler((m.POOLER_RESOLUTION, m.POOLER_RESOLUTION), m.POOLER_SCALES, m.POOLER_SAMPLING_RATIO)
        nxt = cfg.MODEL.BACKBONE.OUT_CHANNELS
        self.blocks = []
        for i, ch in enumerate(m.CONV_LAYERS, 1):
            c = Conv2d(nxt, ch, 3, stride=1, padding=1)
            nn.init.kaiming_normal_(c.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(c.bias, 0)
            self.add_module("mask_fcn%d" % i, c)
            self.blocks.append("mask_fcn%d" % i)
            nxt = ch

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        pre = x
        for i, name in enumerate(self.blocks):
            # (round 6: a layer's output feeds the next 3x3 layer, its input gradient the previous one's data gradient -- both
            # plane-fed launches: the producing epilogues write the planes)
            x = getattr(self, name)(x, relu=True, input_relu=(i > 0), out_rb=(i + 1 < len(self.blocks)), din_rb=(i > 0))
        return x, pre


class MaskRCNNC4Predictor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        nc = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        dim = cfg.MODEL.ROI_MASK_HEAD.CONV_LAYERS[-1]
        self.conv5_mask = ConvTranspose2d(dim, dim, 2, 2, 0)
        self.mask_fcn_logits = Conv2d(dim, nc, 1, 1, 0)
        for name, p in self.named_parameters():
            if "bias" in name:
                nn.init.constant_(p, 0)
            else:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.conv5_mask(x, relu=True, input_relu=True)
        return self.mask_fcn_logits(x, relu=False, input_relu=True)


class MaskRCNNLossComputation(object):
    def __init__(self, proposal_matcher, discretization_size, cfg=None):
        self.proposal_matcher, self.discretization_size, self.cfg = proposal_matcher, discretization_size, cfg

    def prepare_targets(self, proposals, targets):
        """-> labels (cat over images), mask targets (P, M, M) float -- mask_head/loss.py:119-149"""
        labels, mts = [], []
        pm = self.proposal_matcher
        dev = proposals[0].bbox.device
        fused_match = (dev.type == "cuda" and pm.high_threshold == pm.low_threshold and not pm.allow_low_quality_matches
                       and all(len(p) > 0 and len(t) > 0 for p, t in zip(proposals, targets)))
        if fused_match:
            # IoU + Matcher + label lookup of ALL images in one launch (`mmt_match_targets`, the box head's call): with equal
            # thresholds there is no BETWEEN_THRESHOLDS class, so its box-head labels (0 below the threshold, else the matched
            # ground truth's label) are exactly the labels of mask_head/loss.py:119-138
            N = len(proposals)
            A = [len(p) for p in proposals]
            coff, goff = [0], [0]
            for a, t in zip(A, targets):
                coff.append(coff[-1] + a)
                goff.append(goff[-1] + len(t))
            cand = torch.cat([p.bbox for p in proposals], 0) if N > 1 else proposals[0].bbox
            gt = torch.cat([t.bbox.to(dev) for t in targets], 0) if N > 1 else targets[0].bbox.to(dev)
            gl = torch.cat([t.get_field("labels").to(dev) for t in targets], 0) if N > 1 else targets[0].get_field("labels").to(dev)
            mt_all, lab_all, _ = H.match_targets(cand, dev_ints(coff, dev), gt, dev_ints(goff, dev), N,
                                                 pm.high_threshold, pm.low_threshold, False, gt_labels=gl, box_labels=True)
            mi_all = mt_all.clamp(min=0).long()
            pos_all = (lab_all > 0).to(torch.int32)[:, None]
        for i, (p, t) in enumerate(zip(proposals, targets)):
            if fused_match:
                mi, lab = mi_all[coff[i]:coff[i + 1]], lab_all[coff[i]:coff[i + 1]]
            else:
                m = pm(boxlist_iou(t, p))
                mi = m.clamp(min=0)
                lab = t.get_field("labels")[mi].to(torch.int64)
                lab = torch.where(m == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(lab), lab)
            labels.append(lab)
            # every box handed to the mask head is already a positive of the same matcher (mask_head.py:74-78),
            # so `positive_inds` is all of them; non-positives (never produced) would get an all-zero range
            xy, poly_off, inst_rng = t.get_field("masks").packed(p.bbox.device)
            rng = inst_rng[mi] * (pos_all[coff[i]:coff[i + 1]] if fused_match else (lab > 0).to(torch.int32)[:, None])
            mt, ovf = H.polygon_targets(xy, poly_off, rng, p.bbox, self.discretization_size)
            mts.append(mt)
        return torch.cat(labels, 0), torch.cat(mts, 0)

    def __call__(self, proposals, mask_logits, targets):
        labels, mask_targets = self.prepare_targets(proposals, targets)
        if mask_targets.numel() == 0:
            return mask_logits.sum() * 0
        self.last_targets = mask_targets
        return fused.MaskBCEFn.apply(mask_logits, labels, mask_targets)


def make_roi_mask_loss_evaluator(cfg):
    r = cfg.MODEL.ROI_HEADS
    return MaskRCNNLossComputation(Matcher(r.FG_IOU_THRESHOLD, r.BG_IOU_THRESHOLD, allow_low_quality_matches=False),
                                   cfg.MODEL.ROI_MASK_HEAD.RESOLUTION, cfg)


class MaskPostProcessor(nn.Module):
    """mask_head/inference.py:14-65: sigmoid prob of the predicted class, stored per image under 'mask'"""

    def __init__(self, masker=None):
        super().__init__()
        self.masker = masker

    def forward(self, x, boxes):
        labels = torch.cat([b.get_field("labels") for b in boxes])
        per = [len(b) for b in boxes]
        if self.masker is not None:
            segs = self.masker(x, labels, boxes)
        else:
            prob = x.sigmoid()[torch.arange(x.shape[0], device=x.device), labels][:, None]
            segs = prob.split(per, 0)
        out = []
        for s, b in zip(segs, boxes):
            r = BoxList(b.bbox, b.size, "xyxy")
            for f in b.fields():
                r.add_field(f, b.get_field(f))
            r.add_field("mask", s)
            if getattr(b, "count_dev", None) is not None:
                r.count_dev = b.count_dev
            out.append(r)
        return out


class Masker(object):
    """Masker (mask_head/inference.py:209-246) in its only use on the path: the teacher's integral pseudo-mask.
    Returns, per image, an int32 (H, W) map = sum over detections of the pasted binary masks, i.e. exactly
    `t.get_field('mask').sum(0)[0]` of generalized_rcnn.py:129-132, without materialising D canvases."""

    def __init__(self, threshold=0.5, padding=1):
        assert padding == 1
        self.threshold, self.padding = threshold, padding

    def __call__(self, logits, labels, boxes):
        sizes = {b.size for b in boxes}
        if len(sizes) != 1:
            raise NotImplementedError("pseudo-mask paste expects equally sized images in a batch")
        w, h = boxes[0].size
        if all(getattr(b, "count_dev", None) is not None for b in boxes):
            # fixed-capacity detection lists (box_head.py::PostProcessor.forward): rows behind an image's count vote nowhere
            # (image index -1: the paste kernel returns at once)
            cap = len(boxes[0])
            counts = torch.cat([b.count_dev for b in boxes]) if len(boxes) > 1 else boxes[0].count_dev
            rows = torch.arange(cap, device=logits.device, dtype=torch.int32)[None, :]
            ids = torch.arange(len(boxes), device=logits.device, dtype=torch.int32)[:, None]
            img = torch.where(rows < counts[:, None], ids, torch.full_like(ids, -1)).reshape(-1).contiguous()
        else:
            img = torch.cat([torch.full((len(b),), i, dtype=torch.int32, device=logits.device) for i, b in enumerate(boxes)])
        bb = torch.cat([b.bbox for b in boxes], 0)
        seg = H.paste_masks(logits, labels, bb, img, len(boxes), h, w, self.threshold)
        return [IntegralMask(s) for s in seg]


    def forward_single_image(self, masks, boxes):
        """the reference's per-detection form (mask_head/inference.py:221-229), used by the evaluator on predictions whose `mask`
        field is still M x M: masks (n, 1, M, M) probabilities, boxes a BoxList in the target frame -> uint8 (n, 1, H, W)"""
        w, h = boxes.size
        if len(boxes) == 0:
            return torch.zeros((0, 1, h, w), dtype=torch.uint8, device=masks.device)
        return H.paste_mask_stack(masks, boxes.convert("xyxy").bbox, h, w, self.threshold)


class IntegralMask(object):
    """stands in for the (D,1,H,W) uint8 stack of the reference: `.sum(0)[0]` gives the integral mask"""

    def __init__(self, seg):
        self.seg = seg

    def sum(self, dim=0):
        assert dim == 0
        return [self.seg]


def make_roi_mask_post_processor(cfg):
    return MaskPostProcessor(Masker(cfg.MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS_THRESHOLD, 1)
                             if cfg.MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS else None)


def make_roi_mask_generator(cfg):
    return MaskPostProcessor(Masker(cfg.MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS_THRESHOLD, 1))


class ROIMaskHead(nn.Module):
    def __init__(self, cfg, is_student=False):
        super().__init__()
        self.cfg = cfg.clone()
        self.feature_extractor = MaskRCNNFPNFeatureExtractor(cfg)
        self.predictor = MaskRCNNC4Predictor(cfg)
        self.post_processor = make_roi_mask_post_processor(cfg)
        self.mask_generator = make_roi_mask_generator(cfg)
        self.loss_evaluator = make_roi_mask_loss_evaluator(cfg)
        # mask_head.py:49-50 builds the module whatever USE_RELATION says (`if cfg.MODEL.RELATION_MASK:` is a CfgNode),
        # so its keys are in every checkpoint; when unused it never gets a gradient there -> frozen here
        self.use_relation = cfg.MODEL.RELATION_MASK.USE_RELATION
        self.mask_relation_module = MaskRelationRefineNet(cfg, self.predictor)
        if not self.use_relation:
            self.mask_relation_module.requires_grad_(False)
        self.mode = None

    def set_teacher_mode(self, mode):
        self.mode = mode

    def forward(self, features, proposals, targets=None, images=None):
        all_proposals = proposals
        if self.training:
            proposals, _ = keep_only_positive_boxes(proposals)
        x, _ = self.feature_extractor(features, proposals)
        mask_logits = self.predictor(x)
        loss_1 = self.loss_evaluator(proposals, mask_logits, targets) if self.training else None
        if self.use_relation:
            # mask_head.py:98-127: per image, instances sorted per class by objectness, second logits from CIAM
            xr = fused.relu_grad_mask(x)  # x is a fused-ReLU output read by non-fused ops below
            if sum(len(p) for p in proposals) > 0:
                mask_logits, proposals = self.mask_relation_module.forward_batch(xr, mask_logits, proposals)
        if self.training:
            if self.use_relation:
                loss_2 = self.loss_evaluator(proposals, mask_logits, targets)
                loss_1 = 0.5 * (loss_1 + loss_2) if self.cfg.MODEL.RELATION_MASK.DEEP_SUPER else loss_2
            return x, all_proposals, dict(loss_seg=loss_1)
        # D8 of SURVEY.md: the reference hands a tuple to the post-processor here; the only semantics under which
        # the teacher produces a pseudo-mask is the concatenated logits, which is what `mask_logits` already is
        with torch.no_grad():
            if self.mode is None or self.mode == "train":
                result = self.post_processor(mask_logits, proposals)
            else:
                result = self.mask_generator(mask_logits, proposals)
        return x, result, {}


def build_roi_mask_head(cfg, is_student=False):
    return ROIMaskHead(cfg, is_student)
