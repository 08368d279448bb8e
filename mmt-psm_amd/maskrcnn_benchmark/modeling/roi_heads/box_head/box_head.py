"""Box head (reference: modeling/roi_heads/box_head/{box_head,roi_box_feature_extractors,roi_box_predictors,
loss,inference}.py).  Same module tree (feature_extractor.fc6/fc7, predictor.cls_score/bbox_pred,
hint adaptor) and the same three entry points: forward / forward_teacher / forward_student.

Execution: 4-level ROIAlign in one launch -> fc6/fc7 on the fp32 MFMA GEMM with bias+ReLU(+dropout) in the
epilogue -> cls_score and bbox_pred as ONE 15-wide GEMM; PSM loss as two small kernels."""
import os

import torch

from maskrcnn_benchmark.utils.miscellaneous import dev_const, dev_ints
from torch import nn
import torch.nn.functional as F

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import Conv2d, Linear, fused, smooth_l1_loss
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from maskrcnn_benchmark.modeling.poolers import Pooler
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
from maskrcnn_benchmark.utils.miscellaneous import batch_boxlist_hflip




_ARANGE = {}


def _arange(n, dev):
    """cached int32 arange (a constant of the capacity: no launch per call)"""
    t = _ARANGE.get((n, dev))
    if t is None:
        if len(_ARANGE) > 64:
            _ARANGE.clear()
        t = _ARANGE[(n, dev)] = torch.arange(n, device=dev, dtype=torch.int32)
    return t


class MaskRCNNFPNAdaptor(nn.Module):
    """MGD hint adaptors: 5 independent 1x1 convs (roi_box_feature_extractors.py:45-75)"""

    def __init__(self, cfg):
        super().__init__()
        out = 128 if cfg.MT.T_ADAPT is True else 256
        for i in range(1, 6):
            m = Conv2d(256, out, 1, 1, 0)
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)
            self.add_module("adapter_%d" % i, m)

    def forward(self, features_s):
        return [getattr(self, "adapter_%d" % (i + 1))(f) for i, f in enumerate(features_s)]


class FPN2MLPFeatureExtractor(nn.Module):
    """roi_box_feature_extractors.py:77-125.  fc6.weight is held with its 12544 columns in (h, w, c) order --
    the order the NHWC ROIAlign output flattens to -- and converted from/to the reference's (c, h, w) order at
    the state-dict boundary, so checkpoints stay interchangeable."""

    def __init__(self, cfg):
        super().__init__()
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler((res, res), cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES, cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO)
        self.res, self.ch = res, cfg.MODEL.BACKBONE.OUT_CHANNELS
        rep = cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
        self.fc6 = Linear(self.ch * res * res, rep)
        self.fc7 = Linear(rep, rep)
        self.p_drop = cfg.MODEL.ROI_BOX_HEAD.DO
        self.replay = None
        self.generator = None  # see BalancedPositiveNegativeSampler.generator
        self._register_state_dict_hook(self._to_reference_layout)
        self._register_load_state_dict_pre_hook(self._from_reference_layout)

    def _perm(self, w, to_ref):
        O = w.shape[0]
        if to_ref:
            return w.view(O, self.res, self.res, self.ch).permute(0, 3, 1, 2).reshape(O, -1)
        return w.view(O, self.ch, self.res, self.res).permute(0, 2, 3, 1).reshape(O, -1)

    @staticmethod
    def _to_reference_layout(module, sd, prefix, local_metadata):
        k = prefix + "fc6.weight"
        if k in sd:
            sd[k] = module._perm(sd[k], True).contiguous()

    def _from_reference_layout(self, sd, prefix, *args):
        k = prefix + "fc6.weight"
        if k in sd:
            sd[k] = self._perm(sd[k], False).contiguous()

    def forward(self, x, proposals, filp=False, istrain=False):
        pooled = self.pooler(x, proposals)                 # (R, C, 7, 7) NHWC-dense
        x = pooled.permute(0, 2, 3, 1).reshape(pooled.shape[0], -1)  # (R, 7*7*C) view
        fused._carry_stats(pooled, x)                      # (the statistics slot of the pooled tensor: fc6's scale, no reduction pass)
        x = self.fc6(x, relu=True)
        mul = None
        if self.p_drop > 0 and istrain:
            rec = self.replay("dropout") if self.replay is not None else None
            keep = rec.to(x.device) if rec is not None else torch.empty(
                (x.shape[0], self.fc7.out_features), device=x.device).bernoulli_(1 - self.p_drop, generator=self.generator)
            mul = keep / (1 - self.p_drop)
        return self.fc7(x, relu=True, input_relu=True, mul=mul)


class FPNPredictor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        nc, rep = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES, cfg.MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM
        self.cls_score = Linear(rep, nc)
        self.bbox_pred = Linear(rep, nc * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        self.nc = nc

    def forward(self, x, in_mask_scale=1.0):
        w = torch.cat([self.cls_score.weight, self.bbox_pred.weight], 0)
        b = torch.cat([self.cls_score.bias, self.bbox_pred.bias], 0)
        o = fused.linear(x, w, b, False, True, in_mask_scale)
        return o[:, :self.nc], o[:, self.nc:]


def sharpen(p, temp=0.5):
    pt = p ** (1 / temp)
    return (pt / pt.sum(dim=1, keepdim=True)).detach()


class FastRCNNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cfg=None):
        self.proposal_matcher, self.fg_bg_sampler, self.box_coder, self.cfg = proposal_matcher, fg_bg_sampler, box_coder, cfg

    def prepare_targets(self, proposals, targets):
        """box_head/loss.py:38-80 for all images at once (`mmt_match_targets`: IoU + Matcher + labels + encode)"""
        N = len(proposals)
        dev = proposals[0].bbox.device
        for p, t in zip(proposals, targets):
            if len(t) == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            if len(p) == 0:
                raise ValueError("No proposal boxes available for one of the images during training")
        A = [len(p) for p in proposals]
        coff, goff = [0], [0]
        for a, t in zip(A, targets):
            coff.append(coff[-1] + a)
            goff.append(goff[-1] + len(t))
        cand = torch.cat([p.bbox for p in proposals], 0) if N > 1 else proposals[0].bbox
        gt = torch.cat([t.bbox.to(dev) for t in targets], 0) if N > 1 else targets[0].bbox.to(dev)
        gl = torch.cat([t.get_field("labels").to(dev) for t in targets], 0) if N > 1 else targets[0].get_field("labels").to(dev)
        m = self.proposal_matcher
        # data-dependent offsets (proposal counts): through the pinned ring -- a pageable copy would drain the stream first
        _, lab, reg = H.match_targets(cand, dev_ints(coff, dev), gt, dev_ints(goff, dev), N, m.high_threshold,
                                      m.low_threshold, m.allow_low_quality_matches, gt_labels=gl, box_labels=True,
                                      weights=self.box_coder.weights)
        return list(lab.split(A, 0)), list(reg.split(A, 0))

    def subsample_fixed(self, proposals, targets):
        """box_head/loss.py:82-116 on FIXED-CAPACITY proposal lists (rpn.py::select with `fixed_capacity`: every image has `cap`
        rows, the rows behind its device-side count are zero boxes) without a host read-back (SURVEY f-2, round 6).  Rows behind the
        count are labelled -1 -- the sampler's "never" -- so the sampled sets are those of the sliced lists; every image then gets
        exactly BATCH_SIZE_PER_IMAGE rows, the sampled ones in ascending order like `nonzero`, followed -- only when the image had
        fewer candidates than that -- by rows labelled -1 that the losses skip (`mmt_box_loss` normalises by the rows it counts).
        The number of positives per image, which sizes the mask head's input, travels through a pinned buffer behind an event
        (`n_pos_async`): whoever needs it waits there, after the box head's launches have been queued."""
        N, cap = len(proposals), len(proposals[0])
        dev = proposals[0].bbox.device
        for t in targets:
            if len(t) == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
        goff = [0]
        for t in targets:
            goff.append(goff[-1] + len(t))
        cand = torch.cat([p.bbox for p in proposals], 0) if N > 1 else proposals[0].bbox
        gt = torch.cat([t.bbox.to(dev) for t in targets], 0) if N > 1 else targets[0].bbox.to(dev)
        gl = torch.cat([t.get_field("labels").to(dev) for t in targets], 0) if N > 1 else targets[0].get_field("labels").to(dev)
        m = self.proposal_matcher
        _, lab, reg = H.match_targets(cand, dev_const([i * cap for i in range(N + 1)], torch.int32, dev), gt, dev_ints(goff, dev), N,
                                      m.high_threshold, m.low_threshold, m.allow_low_quality_matches, gt_labels=gl, box_labels=True,
                                      weights=self.box_coder.weights)
        counts = torch.cat([p.count_dev for p in proposals]) if N > 1 else proposals[0].count_dev
        rows = _arange(cap, dev)
        lab = torch.where((rows[None, :] < counts[:, None]).reshape(-1), lab, torch.full_like(lab, -1))
        labels = list(lab.split(cap, 0))
        pos, neg = self.fg_bg_sampler(labels, tag="roi_sampler")
        B = self.fg_bg_sampler.batch_size_per_image
        npos = torch.stack([p_.sum() for p_ in pos])
        pin = getattr(self, "_npos_pin", None)
        if pin is None or pin.numel() < N:
            pin = self._npos_pin = torch.empty((max(N, 8),), dtype=torch.int64).pin_memory()
        pin[:N].copy_(npos, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        out = []
        regs = reg.split(cap, 0)
        for i, (p, lb, rg, pm_, nm_) in enumerate(zip(proposals, labels, regs, pos, neg)):
            idx = torch.nonzero_static(pm_ | nm_, size=B, fill_value=-1).squeeze(1)
            ok = idx >= 0
            idx = idx.clamp(min=0)
            q = BoxList(p.bbox[idx], p.size, p.mode)
            for k, v in p.extra_fields.items():
                q.add_field(k, v[idx])
            q.add_field("labels", torch.where(ok, lb[idx], torch.full_like(idx, -1).to(lb.dtype)))
            q.add_field("regression_targets", rg[idx])
            q.n_pos_async = (pin, ev, i)
            out.append(q)
        self._proposals = out
        return out

    def subsample(self, proposals, targets):
        """box_head/loss.py:82-116"""
        if all(getattr(p, "count_dev", None) is not None for p in proposals):
            return self.subsample_fixed(proposals, targets)
        labels, regs = self.prepare_targets(proposals, targets)
        pos, neg = self.fg_bg_sampler(labels, tag="roi_sampler")
        out = []
        # ONE host transfer for the whole batch (sampled and positive counts per image) instead of one blocking nonzero per
        # image; the positive counts travel with the proposals so the mask head needs no sync of its own
        both = [pm | nm for pm, nm in zip(pos, neg)]
        cnts = torch.stack([m.sum() for m in both] + [m.sum() for m in pos]).tolist()
        N = len(proposals)
        for i, (p, lab, rg, m) in enumerate(zip(proposals, labels, regs, both)):
            p.add_field("labels", lab)
            p.add_field("regression_targets", rg)
            q = p[torch.nonzero_static(m, size=cnts[i]).squeeze(1)]
            q.n_pos = cnts[N + i]
            out.append(q)
        self._proposals = out
        return out

    def __call__(self, class_logits, box_regression):
        """box_head/loss.py:118-162"""
        class_logits, box_regression = torch.cat(class_logits, 0), torch.cat(box_regression, 0)
        props = self._proposals
        labels = torch.cat([p.get_field("labels") for p in props], 0)
        regt = torch.cat([p.get_field("regression_targets") for p in props], 0)
        # one launch (csrc/losses.hip: mmt_box_loss); its tensor formulation is the checker in tests/test_hip_kernels.py.  Rows
        # labelled -1 (fixed-capacity lists: subsample_fixed) are not rows of the reference's batch: skipped, not counted
        fixed = any(getattr(p, "n_pos_async", None) is not None for p in props)
        return fused.BoxLossFn.apply(class_logits, box_regression, labels, regt, (labels >= 0).sum() if fixed else None)

    def evaluatePSM(self, class_logits, class_logits_t, proposals):
        """box_head/loss.py:164-237,267-287: hard-negative mining by teacher disagreement + sharpened soft CE"""
        cfg = self.cfg
        labels = torch.cat([p.get_field("labels") for p in proposals], 0)
        teacher = torch.stack([t.detach() for t in class_logits_t]).contiguous()  # (K, R, NC)
        typ = cfg.MT.CLS_LOSS_TYPE
        kind = {"bce": 0, "ce": 0, "kl": 1, "mse": 2}.get(typ)
        if kind is None:
            raise NotImplementedError("MT.CLS_LOSS_TYPE=%s: the MI355X path implements 'bce'/'ce', 'kl' and 'mse'" % typ)
        pos, neg = labels > 0, labels == 0
        n_pos, n_neg = pos.sum(), neg.sum()
        if cfg.MT.RANK_FILTER > 0:
            if cfg.MT.HARD_NEG:
                v = H.psm_variance(teacher, use_softmax=(typ == "bce"))
            else:
                v = torch.rand(labels.shape, device=labels.device, generator=self.fg_bg_sampler.generator)
            vn = torch.where(neg, v, torch.full_like(v, -1.0))
            order = torch.argsort(vn, descending=True, stable=True)
            rank = torch.empty_like(order)
            rank[order] = torch.arange(order.numel(), device=order.device)
            n_keep = torch.minimum(n_neg, n_pos // 2)
            keep_neg = neg & (rank < n_keep)
            wneg = cfg.MT.CLS_BALANCE_WEIGHT if (cfg.MT.HARD_NEG and kind == 0) else 1.0
            roww = pos.to(torch.float32) + keep_neg.to(torch.float32) * wneg
            S = (n_pos + n_keep).to(torch.float32)
        else:
            # no filtering: every row, weight 1.  For 'bce' the reference hands cls_loss the mean of the per-view softmax
            # PROBABILITIES (m_logit_t of _mean_var_logits, box_head/loss.py:164-173,229) and cls_loss softmaxes that again
            # (:281); reproduced as written: the kernel's "mean over views, then softmax" over ONE pre-averaged view
            if typ == "bce":
                teacher = torch.softmax(teacher, dim=2).mean(0, keepdim=True).contiguous()
            roww = (labels >= 0).to(torch.float32)   # (rows labelled -1: padding of a fixed-capacity list, not rows of the batch)
            S = roww.sum()
        nc = teacher.shape[2]
        norm = 1.0 / (S * (3.0 if kind == 0 else float(nc)))
        losses = [fused.PSMLossFn.apply(cl, teacher, roww, norm, cfg.MT.TEMP, 1 if cfg.MT.SHARPEN else 0, kind)
                  for cl in class_logits]
        return torch.mean(torch.stack(losses), dim=0)


def make_roi_box_loss_evaluator(cfg):
    r = cfg.MODEL.ROI_HEADS
    return FastRCNNLossComputation(Matcher(r.FG_IOU_THRESHOLD, r.BG_IOU_THRESHOLD, allow_low_quality_matches=False),
                                   BalancedPositiveNegativeSampler(r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION),
                                   BoxCoder(weights=r.BBOX_REG_WEIGHTS), cfg=cfg)


class PostProcessor(nn.Module):
    """box_head/inference.py:11-145: softmax, decode, clip, per-class score threshold + NMS, keep top-k.
    All (image, class) NMS problems of the batch go through one `mmt_nms_batched` call."""

    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None, cfg=None):
        super().__init__()
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(10., 10., 5., 5.))

    def forward(self, x, boxes):
        class_logits, box_regression = x
        prob = F.softmax(class_logits, -1)
        per = [len(b) for b in boxes]
        dev = prob.device
        if all(getattr(b, "count_dev", None) is not None for b in boxes):
            # fixed-capacity proposal lists (rpn.py::select): rows behind an image's count are not proposals -- probability 0
            # keeps them below every score threshold, the valid rows keep their values (x 1.0) and their order
            assert len(set(per)) == 1
            counts = torch.cat([b.count_dev for b in boxes]) if len(boxes) > 1 else boxes[0].count_dev
            valid = torch.arange(per[0], device=dev, dtype=torch.int32)[None, :] < counts[:, None]
            prob = prob * valid.reshape(-1, 1).to(prob.dtype)
        cat = torch.cat([b.bbox for b in boxes], 0)
        offs_rows = [0]
        for n_ in per:
            offs_rows.append(offs_rows[-1] + n_)
        # decode (10,10,5,5) + clip_to_image of every (proposal, class): one launch
        dec = H.box_decode(box_regression.reshape(sum(per), -1), cat, self.box_coder.weights, self.box_coder.bbox_xform_clip,
                           dev_const(offs_rows, torch.int32, dev),
                           dev_const([[b.size[0] - 1, b.size[1] - 1] for b in boxes], torch.float32, dev))
        # threshold / stable sort / NMS / ascending-row order / DETECTIONS_PER_IMG cut of every (image, class) on the
        # device (mmt_det_postprocess: five launches), ONE read-back (the detection counts) for the BoxList sizes.  The tensor
        # formulation it replaced is its checker (tests/tensor_formulations.py::det_filter_results)
        defer = getattr(self, "defer_counts", False)
        out = H.det_postprocess(prob, dec, per, self.score_thresh, self.nms, self.detections_per_img, zero_tails=defer)
        if out is None:
            raise RuntimeError("PostProcessor: at most 2048 proposals per image and 64 classes (mmt_det_postprocess is the only "
                               "implementation)")
        ob, os_, ol, oc = out
        results = []
        if defer:
            # SURVEY f-2 (the teacher's coarse inference): detection lists at the fixed capacity DETECTIONS_PER_IMG with the count
            # as a device scalar; rows behind it are zero boxes with label 0.  The mask head runs on all rows and its paste skips
            # them (mask_head.py::Masker); `resolve_counts` slices the lists once that work is queued.
            k = min(self.detections_per_img, ob.shape[1]) if self.detections_per_img > 0 else ob.shape[1]
            for i, b in enumerate(boxes):
                r = BoxList(ob[i, :k], b.size, "xyxy")
                r.add_field("scores", os_[i, :k])
                r.add_field("objectness", os_[i, :k])
                r.add_field("labels", ol[i, :k])
                r.count_dev = oc[i:i + 1]
                results.append(r)
            return results
        for i, (b, n) in enumerate(zip(boxes, oc.tolist())):
            r = BoxList(ob[i, :n], b.size, "xyxy")
            r.add_field("scores", os_[i, :n])
            r.add_field("objectness", os_[i, :n])
            r.add_field("labels", ol[i, :n])
            results.append(r)
        return results


def resolve_counts(boxlists):
    """fixed-capacity lists (`count_dev`) -> the sliced lists of the reference; ONE read-back for all of them"""
    counts = (torch.cat([b.count_dev for b in boxlists]) if len(boxlists) > 1 else boxlists[0].count_dev).tolist()
    out = []
    for b, n in zip(boxlists, counts):
        r = BoxList(b.bbox[:n], b.size, b.mode)
        for f in b.fields():
            v = b.get_field(f)
            r.add_field(f, v[:n] if torch.is_tensor(v) else v)
        out.append(r)
    return out


def make_roi_box_post_processor(cfg):
    r = cfg.MODEL.ROI_HEADS
    return PostProcessor(r.SCORE_THRESH, r.NMS, r.DETECTIONS_PER_IMG, BoxCoder(weights=r.BBOX_REG_WEIGHTS), cfg=cfg)


class ROIBoxHead(nn.Module):
    def __init__(self, cfg, relation=True):
        super().__init__()
        self.feature_extractor = FPN2MLPFeatureExtractor(cfg)
        self.predictor = FPNPredictor(cfg)
        self.post_processor = make_roi_box_post_processor(cfg)
        self.loss_evaluator = make_roi_box_loss_evaluator(cfg)
        self.use_realation_nms = relation and cfg.MODEL.RELATION_NMS.USE_RELATION_NMS  # (sic) box_head.py:24
        self.cfg = cfg
        self.mode = None

    def set_teacher_mode(self, mode):
        self.mode = mode

    def _scale(self, istrain):
        p = self.feature_extractor.p_drop
        return 1.0 / (1 - p) if (p > 0 and istrain) else 1.0

    def forward(self, features, proposals, targets=None):
        if self.training:
            with torch.no_grad():
                proposals = self.loss_evaluator.subsample(proposals, targets)
        x = self.feature_extractor(features, proposals, istrain=self.training)
        class_logits, box_regression = self.predictor(x, self._scale(self.training))
        if not self.training:
            if not self.use_realation_nms:  # with IR-Net the relation module replaces score-threshold + greedy NMS
                with torch.no_grad():
                    proposals = self.post_processor((class_logits, box_regression), proposals)
            return x, proposals, {}, class_logits, box_regression
        lc, lb = self.loss_evaluator([class_logits], [box_regression])
        return x, proposals, dict(loss_classifier=lc, loss_box_reg=lb), class_logits, box_regression

    def _forward_single(self, proposals, targets, feats_list, istrain=False):
        if targets is not None:
            proposals = self.loss_evaluator.subsample(proposals, targets)
        proposals_b = batch_boxlist_hflip(proposals)
        feats, logits, regs = [], [], []
        bp = getattr(self, "batched_pyramid", None)
        if bp is not None and len(feats_list) == bp[2] and feats_list[0][0].data_ptr() == bp[0][0].data_ptr():
            # all views came out of one batched backbone pass: ONE 4-level ROIAlign + fc6/fc7/predictor over the
            # ROIs of every view (image index = view * n + image), then split per view
            pyr, n, nv = bp
            boxes = []
            for i in range(nv):
                boxes += list(proposals if i % 2 == 0 else proposals_b)
            x = self.feature_extractor(pyr, boxes, istrain=istrain)
            cl, br = self.predictor(x, self._scale(istrain))
            R = x.shape[0] // nv
            return (list(x.split(R, 0)), list(cl.split(R, 0)), list(br.split(R, 0)), proposals)
        for i, feat in enumerate(feats_list):
            x = self.feature_extractor(feat, proposals if i % 2 == 0 else proposals_b, istrain=istrain)
            cl, br = self.predictor(x, self._scale(istrain))
            feats.append(x)
            logits.append(cl)
            regs.append(br)
        return feats, logits, regs, proposals

    def forward_teacher(self, feature_tuple, proposals, targets):
        feats, logits, regs, proposals = self._forward_single(proposals, targets, feature_tuple, istrain=False)
        return feats, proposals, {}, logits, regs

    def forward_student(self, features, proposals, class_logits_t):
        _, logits, _, _ = self._forward_single(proposals, targets=None, feats_list=features, istrain=True)
        if self.cfg.MT.CLS_LOSS > 0:
            cls_loss = self.loss_evaluator.evaluatePSM(logits, class_logits_t, proposals)
        else:
            cls_loss = torch.zeros((1,), device=logits[0].device)
        return dict(mt_classifier=cls_loss)


def build_roi_box_head(cfg, relation=True):
    return ROIBoxHead(cfg, relation=relation)
