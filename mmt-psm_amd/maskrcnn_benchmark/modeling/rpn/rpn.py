"""RPN (reference: modeling/rpn/rpn.py:14-221, rpn/inference.py:17-272, rpn/loss.py:22-214).

API kept: RPNModule.forward(images, features, targets) -> (list[BoxList], losses),
RPNModule.forward_teacher(...), set_teacher_mode, `rpn.head.{conv,cls_logits,bbox_pred}` parameters,
`rpn.anchor_generator.cell_anchors.*` buffers.

Execution: head = fused 3x3+bias+ReLU conv followed by ONE 1x1 GEMM producing
objectness and the 4A deltas together (15 channels), all NHWC so the (N,H,W,A[,4]) order the post-processor
wants is the memory order.  Post-processing is batched over images: per level one top-k / gather / decode /
clip over the whole batch, then a SINGLE `mmt_nms_batched` launch pair over all (image, level) segments with
the greedy sweep on the device (the reference does 5*N separate NMS calls each with a D2H mask copy).
"""
import os

import torch

from maskrcnn_benchmark.utils.miscellaneous import dev_const
from torch import nn
import torch.nn.functional as F

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import Conv2d, fused, smooth_l1_loss
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.boxlist_ops import box_iou_tensor
from .anchor_generator import make_anchor_generator


class RPNHead(nn.Module):
    def __init__(self, cfg, in_channels, num_anchors):
        super().__init__()
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self.num_anchors = num_anchors

    def forward(self, x):
        """-> (logits [(N,A,H,W)], bbox_reg [(N,4A,H,W)]) as NHWC-dense views of one 5A-channel tensor"""
        A = self.num_anchors
        w = torch.cat([self.cls_logits.weight, self.bbox_pred.weight], 0).contiguous(memory_format=torch.channels_last)
        b = torch.cat([self.cls_logits.bias, self.bbox_pred.bias], 0)
        logits, regs = [], []
        from maskrcnn_benchmark.layers import fused
        for f in x:
            t = self.conv(f, relu=True)
            o = fused.conv(t, w, b, 1, 0, False, True)
            logits.append(o[:, :A])
            regs.append(o[:, A:])
        return logits, regs


def _flat(obj, reg):
    """(N,A,H,W),(N,4A,H,W) NHWC-dense views -> (N, HWA), (N, HWA, 4) without copies when possible"""
    N, A, Hh, Ww = obj.shape
    return obj.permute(0, 2, 3, 1).reshape(N, -1), reg.permute(0, 2, 3, 1).reshape(N, -1, 4)


class RPNPostProcessor(nn.Module):
    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder=None, fpn_post_nms_top_n=None,
                 is_teacher=False):
        super().__init__()
        self.pre_nms_top_n, self.post_nms_top_n = pre_nms_top_n, post_nms_top_n
        self.nms_thresh, self.min_size = nms_thresh, min_size
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.fpn_post_nms_top_n = post_nms_top_n if fpn_post_nms_top_n is None else fpn_post_nms_top_n
        self.is_teacher = is_teacher

    def forward(self, anchors, objectness, box_regression, targets=None, shared=None, between=None):
        """anchors: list[image] of list[level] BoxList -> list[BoxList] (rpn/inference.py:139-172).
        `shared` (dict or None): the teacher runs its TEST-config and TRAIN-config selectors on the same head outputs
        (generalized_rcnn.py:126,146); decode + NMS are then done once on the larger pre-NMS top-k and each selector
        picks its own prefix -- greedy NMS on a score-sorted list restricted to a prefix IS the NMS of the prefix."""
        pre = self.pre_nms_top_n if shared is None else max(self.pre_nms_top_n, shared.get("pre", 0))
        key = "cands%d" % pre
        if shared is not None and key in shared:
            c = shared[key]
        else:
            c = self.compute_candidates(anchors, objectness, box_regression, pre)
            if shared is not None:
                shared[key] = c
        return self.select(c, targets, between)

    @staticmethod
    def _fused_head(o, r):
        """(N,A,H,W) logits and (N,4A,H,W) deltas -> the (N,5A,H,W) NHWC tensor they are the two channel ranges of (RPNHead
        produces exactly that: zero-copy), or a fused copy when they come from elsewhere"""
        N, A, Hh, Ww = o.shape
        C = 5 * A
        st = (Hh * Ww * C, 1, Ww * C, C)
        if o.stride() == st and r.stride() == st and r.data_ptr() == o.data_ptr() + 4 * A:
            return torch.as_strided(o, (N, C, Hh, Ww), st)
        return torch.cat([o, r], 1).contiguous(memory_format=torch.channels_last)

    def compute_candidates(self, anchors, objectness, box_regression, pre_n):
        """rpn/inference.py:86-135 for every (image, level): top-k of the objectness (`mmt_rpn_topk`: all levels and images in
        five launches, on the logits -- ordered like their sigmoid), then ONE launch for gather + sigmoid + decode + clip of all levels
        (`mmt_rpn_gather_decode`) and one `mmt_nms_batched` launch pair over all segments"""
        N, L = len(anchors), len(objectness)
        dev = objectness[0].device
        sizes = [a[0].size for a in anchors]  # (W,H) per image
        lim = dev_const([[s[0] - 1, s[1] - 1] for s in sizes], torch.float32, dev)
        A = objectness[0].shape[1]
        heads, topks, ks = [], [], []
        for lvl in range(L):
            head = self._fused_head(objectness[lvl].detach(), box_regression[lvl].detach())
            heads.append(head)
            ks.append(min(pre_n, head.shape[2] * head.shape[3] * A))
        if max(ks) <= 2048 and heads[0].dtype == torch.float32:
            topks = H.rpn_topk(heads, ks, A)   # every (image, level) in five launches
        else:                                  # PRE_NMS_TOP_N beyond the kernel's candidate list: the library top-k per level
            topks = [h[:, :A].permute(0, 2, 3, 1).reshape(N, -1).topk(k, dim=1, sorted=True)[1] for h, k in zip(heads, ks)]
        kmax = max(ks)
        boxes, scores, idx, reg, offs = H.rpn_gather_decode(heads, [anchors[0][lvl].bbox for lvl in range(L)], topks, A,
                                                            self.box_coder.bbox_xform_clip, lim)
        if self.min_size > 0:
            # remove_small_boxes BEFORE the NMS (rpn/inference.py:124-129): a removed box must neither suppress anything nor
            # take a post-NMS slot.  Fixed shapes: it keeps its position in the segment but is moved far outside the image
            # (IoU 0 with every real box) and gets score -1, which the selection excludes before it counts the slots.
            ws = boxes[..., 2] - boxes[..., 0] + 1
            hs = boxes[..., 3] - boxes[..., 1] + 1
            ok = (ws >= self.min_size) & (hs >= self.min_size)
            scores = torch.where(ok, scores, torch.full_like(scores, -1.0))
            boxes = torch.where(ok[..., None], boxes, torch.full_like(boxes, -1.0e6))
        sumk = offs[-1]
        seg = [n * sumk + offs[l] for n in range(N) for l in range(L)] + [N * sumk]
        keep, cnt = H.nms_batched(boxes.view(-1, 4), dev_const(seg, torch.int32, dev), kmax, self.nms_thresh)
        return dict(N=N, L=L, ks=ks, kmax=kmax, sizes=sizes, boxes=boxes, scores=scores, idx=idx, reg=reg, offs=offs,
                    keep=keep, cnt=cnt, dev=dev)

    def select(self, c, targets=None, between=None):
        """rpn/inference.py:130-135 (per-level POST_NMS_TOP_N), :216-243 (FPN_POST_NMS_TOP_N over the batch in training,
        per image in score order otherwise), :55-76 (ground-truth boxes appended in training): one launch
        (`mmt_rpn_post_select`) into fixed-capacity tensors; only the per-image counts cross to the host."""
        N, L, ks, kmax, dev = c["N"], c["L"], c["ks"], c["kmax"], c["dev"]
        sizes = c["sizes"]
        training = self.training
        own_pre = [min(self.pre_nms_top_n, k) for k in ks]
        per_seg = min(self.post_nms_top_n, kmax) if self.post_nms_top_n > 0 else kmax
        if L > 1:
            fpn = min(self.fpn_post_nms_top_n, N * L * per_seg if training else L * per_seg)
        else:  # single feature map: no cut over levels (rpn/inference.py:165-166)
            fpn = N * per_seg if training else per_seg
        gt = gt_off = None
        max_gt = 0
        if training and targets is not None:  # add_gt_proposals
            goff = [0]
            for t in targets:
                goff.append(goff[-1] + len(t))
            max_gt = max(len(t) for t in targets)
            gt = torch.cat([t.bbox.to(dev) for t in targets], 0).contiguous() if N > 1 else targets[0].bbox.to(dev).contiguous()
            gt_off = dev_const(goff, torch.int32, dev)
        cap = min(fpn, L * per_seg) + max_gt
        ob, osc, oi, orr, ol, oc = H.rpn_post_select(c["boxes"], c["scores"], c["idx"], c["reg"], c["keep"], c["cnt"],
                                                     c["offs"], own_pre, self.post_nms_top_n, fpn, training, cap, gt, gt_off,
                                                     min_size_filter=self.min_size > 0,
                                                     zero_tails=getattr(self, "fixed_capacity", False))
        if getattr(self, "fixed_capacity", False):
            # SURVEY f-2: no read-back -- every image keeps all `cap` rows, the rows behind its count are zero boxes, and the count
            # travels as a device scalar (`count_dev`).  Consumers: the box head's inference post-processor gives those rows
            # probability 0 (they can never pass its score threshold: box_head.py::PostProcessor.forward); round 6, the TRAINING
            # lists of the student and of the teacher's train-config selector: the box head's sampler labels them -1 = never
            # sampled (box_head.py::FastRCNNLossComputation.subsample_fixed), so the sampled sets are those of the sliced lists
            if between is not None:
                self.between_result = between()
            out = []
            for n in range(N):
                b = BoxList(ob[n], sizes[n], "xyxy")
                b.add_field("objectness", osc[n])
                if self.is_teacher:
                    b.add_field("box_reg", orr[n])
                    b.add_field("rpn_topk", oi[n])
                    b.add_field("rpn_ancher_level", ol[n].to(torch.int64))
                b.count_dev = oc[n:n + 1]
                out.append(b)
            return out
        # the one host sync of the proposal pipeline: the per-image counts, through a pinned buffer and an event, so that
        # whatever `between` enqueues (the RPN losses) keeps the GPU busy while the host is released
        counts = self._counts_to_host(oc, between)
        out = []
        for n in range(N):
            k = counts[n]
            b = BoxList(ob[n, :k], sizes[n], "xyxy")
            b.add_field("objectness", osc[n, :k])
            if self.is_teacher:
                b.add_field("box_reg", orr[n, :k])
                b.add_field("rpn_topk", oi[n, :k])
                b.add_field("rpn_ancher_level", ol[n, :k].to(torch.int64))
            out.append(b)
        return out

    def _counts_to_host(self, cnt, between):
        """device int64 vector -> python list; `between()` runs after the copy is queued and before the host waits"""
        if not cnt.is_cuda:
            if between is not None:
                self.between_result = between()
            return cnt.tolist()
        pin = getattr(self, "_pin", None)
        if pin is None or pin.numel() < cnt.numel() or pin.dtype != cnt.dtype:
            pin = self._pin = torch.empty(max(16, cnt.numel()), dtype=cnt.dtype).pin_memory()
        pin[:cnt.numel()].copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if between is not None:
            self.between_result = between()
        ev.synchronize()
        return pin[:cnt.numel()].tolist()


def make_rpn_postprocessor(config, rpn_box_coder, is_train, is_teacher=False):
    r = config.MODEL.RPN
    return RPNPostProcessor(
        pre_nms_top_n=r.PRE_NMS_TOP_N_TRAIN if is_train else r.PRE_NMS_TOP_N_TEST,
        post_nms_top_n=r.POST_NMS_TOP_N_TRAIN if is_train else r.POST_NMS_TOP_N_TEST,
        nms_thresh=r.NMS_THRESH, min_size=r.MIN_SIZE, box_coder=rpn_box_coder,
        fpn_post_nms_top_n=r.FPN_POST_NMS_TOP_N_TRAIN if is_train else r.FPN_POST_NMS_TOP_N_TEST,
        is_teacher=is_teacher)


class RPNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cfg=None):
        self.proposal_matcher, self.fg_bg_sampler, self.box_coder = proposal_matcher, fg_bg_sampler, box_coder

    def prepare_targets(self, anchors, targets):
        """anchors: list[image] of (bbox (A,4), visibility (A,), area (A,)) ; rpn/loss.py:56-83.
        IoU, Matcher (with low-quality matches), the label rules and BoxCoder.encode for all images: ONE
        `mmt_match_targets` call (two launches) instead of ~45 tensor launches per image over [G x A] matrices."""
        N = len(targets)
        dev = anchors[0][0].device
        for t in targets:
            if len(t) == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
        ab, vis = anchors[0][0], anchors[0][1]
        shared = all(a[0].data_ptr() == ab.data_ptr() for a in anchors)  # one anchor grid for equally sized images
        if not shared:
            ab = torch.cat([a[0] for a in anchors], 0)
            vis = torch.cat([a[1] for a in anchors], 0)
        A = [a[0].shape[0] for a in anchors]
        coff, goff = [0], [0]
        for a, t in zip(A, targets):
            coff.append(coff[-1] + a)
            goff.append(goff[-1] + len(t))
        gt = torch.cat([t.bbox.to(dev) for t in targets], 0) if N > 1 else targets[0].bbox.to(dev)
        m = self.proposal_matcher
        _, lab, reg = H.match_targets(ab, dev_const(coff, torch.int32, dev), gt, dev_const(goff, torch.int32, dev), N,
                                      m.high_threshold, m.low_threshold, m.allow_low_quality_matches, visible=vis,
                                      shared_cand=shared, rpn_labels=True, weights=self.box_coder.weights)
        return list(lab.split(A, 0)), list(reg.split(A, 0))

    def _cat_anchors(self, anchors):
        """per image (all-level anchors, visibility, area): constants of the anchor grid, cached on its device addresses"""
        cache = self.__dict__.setdefault("_anchor_cache", {})
        out = []
        for per_img in anchors:
            key = tuple((a.bbox.data_ptr(), a.get_field("visibility").data_ptr()) for a in per_img)
            ent = cache.get(key)
            if ent is None:
                if len(cache) > 64:
                    cache.clear()
                b = torch.cat([a.bbox for a in per_img], 0)
                v = torch.cat([a.get_field("visibility") for a in per_img], 0)
                ent = cache[key] = (b, v, (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1),
                                    [a.bbox for a in per_img], [a.get_field("visibility") for a in per_img])  # keep-alive
            out.append(ent[:3])
        return out

    def teacher_sample_selection(self, anchors, objectness, box_regression, targets):
        """rpn/loss.py:85-136 -- its outputs are unused downstream (generalized_rcnn.py:146-148); only the RNG
        draw matters for seed-for-seed parity, so the device sampler is exercised and nothing else."""
        rp = self.fg_bg_sampler.replay
        if rp is not None:
            rp("teacher_rpn_sampler")  # consume the recorded draw; the labels/samples themselves are never used

    def __call__(self, anchors, objectness, box_regression, targets):
        labels, regt = self.prepare_targets(self._cat_anchors(anchors), targets)
        pos, neg = self.fg_bg_sampler(labels, tag="rpn_sampler")
        pos, neg = torch.cat(pos, 0), torch.cat(neg, 0)
        of, rf = [], []
        for o, r in zip(objectness, box_regression):
            o2, r2 = _flat(o, r)
            of.append(o2)
            rf.append(r2)
        obj = torch.cat(of, 1).reshape(-1)
        reg = torch.cat(rf, 1).reshape(-1, 4)
        labels, regt = torch.cat(labels, 0), torch.cat(regt, 0)
        # two launches (csrc/losses.hip: mmt_rpn_loss); the tensor formulation of rpn/loss.py:183-194 is its checker in
        # tests/test_hip_kernels.py
        return fused.RPNLossFn.apply(obj, reg, labels, regt, pos, neg, 1.0 / 9)


def make_rpn_loss_evaluator(cfg, box_coder):
    r = cfg.MODEL.RPN
    return RPNLossComputation(Matcher(r.FG_IOU_THRESHOLD, r.BG_IOU_THRESHOLD, allow_low_quality_matches=True),
                              BalancedPositiveNegativeSampler(r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION),
                              box_coder, cfg)


class RPNModule(nn.Module):
    def __init__(self, cfg, is_teacher=False):
        super().__init__()
        self.mode = None
        self.cfg = cfg.clone()
        self.anchor_generator = make_anchor_generator(cfg)
        self.head = RPNHead(cfg, cfg.MODEL.BACKBONE.OUT_CHANNELS, self.anchor_generator.num_anchors_per_location()[0])
        coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.box_selector_train = make_rpn_postprocessor(cfg, coder, is_train=True, is_teacher=is_teacher)
        self.box_selector_test = make_rpn_postprocessor(cfg, coder, is_train=False, is_teacher=is_teacher)
        self.loss_evaluator = make_rpn_loss_evaluator(cfg, coder)

    def set_teacher_mode(self, mode):
        self.mode = mode

    def _head(self, features):
        """head outputs, shared between the coarse inference and forward_teacher of ONE forward_teacher call (both
        run on pyramid 0; the reference recomputes them, generalized_rcnn.py:126 vs :146)"""
        sh = getattr(self, "shared", None)
        if sh is not None and sh.get("feat_id") == id(features[0]):
            return sh["head"]
        out = self.head(features)
        if sh is not None:
            sh["feat_id"], sh["head"], sh["keepalive"] = id(features[0]), out, features[0]
        return out

    def forward(self, images, features, targets=None):
        objectness, rpn_box_regression = self._head(features)
        anchors = self.anchor_generator(images, features)
        if self.training or self.mode == "train":
            grad_on = torch.is_grad_enabled()

            def losses():
                with torch.set_grad_enabled(grad_on):
                    return self.loss_evaluator(anchors, objectness, rpn_box_regression, targets)

            with torch.no_grad():  # the losses are enqueued inside the selector, between its count copy and its host wait
                boxes = self.box_selector_train(anchors, objectness, rpn_box_regression, targets, between=losses)
            lo, lb = self.box_selector_train.between_result
            self.box_selector_train.between_result = None
            return boxes, {"loss_objectness": lo, "loss_rpn_box_reg": lb}
        with torch.no_grad():
            boxes = self.box_selector_test(anchors, objectness, rpn_box_regression, shared=getattr(self, "shared", None))
        return boxes, {}

    def forward_teacher(self, images, features, targets=None):
        """rpn/rpn.py:146-177 (single pyramid branch; FFI imitation boxes are a compared method, off)"""
        objectness, rpn_box_regression = self._head(features)
        anchors = self.anchor_generator(images, features)
        with torch.no_grad():
            boxes = self.box_selector_train(anchors, objectness, rpn_box_regression, targets,
                                            shared=getattr(self, "shared", None))
            self.loss_evaluator.teacher_sample_selection(anchors, objectness, rpn_box_regression, targets)
        return None, None, None, None, boxes, {}, None


def build_rpn(cfg, is_teacher=False):
    return RPNModule(cfg, is_teacher=is_teacher)
