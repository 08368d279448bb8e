"""GeneralizedRCNN with the mean-teacher entry points (reference: modeling/detector/generalized_rcnn.py:17-282).

forward(images, targets)            supervised student step -> loss dict / eval detections
forward_teacher(images)             coarse inference -> pseudo labels + masks; K-aug x flip pyramids; per-view logits
forward_student(images, result_t)   MGD (mt_fg_loss) + PSM (mt_classifier)

Same attribute tree (backbone, rpn, box_heads, mask_heads, relation_nms, hint_adaptor), same dict keys.  The IR-Net
branches (relation_nms here, mask relation inside the mask head) follow generalized_rcnn.py:62-96."""
import torch
from torch import nn

from maskrcnn_benchmark.layers import fused
from maskrcnn_benchmark.structures.image_list import to_image_list
from ..backbone import build_backbone
from ..rpn.rpn import build_rpn
from ..roi_heads.roi_heads import box_roi_heads, mask_roi_heads
from ..roi_heads.box_head.box_head import MaskRCNNFPNAdaptor
from ..relation.relation_module import DuplicationRemovalNetwork


class ImageListView(object):
    """an ImageList snapshot that is not affected by the in-place hflip() of extract_aug_feat"""

    def __init__(self, il):
        self.tensors, self.image_sizes = il.tensors, list(il.image_sizes)


# SURVEY f-2: the teacher's coarse inference keeps its proposal and detection lists at fixed capacity with device counts (no
# read-back before the box head, the detection counts read back behind the queued mask head).  False = sliced lists throughout
# (tests/test_model_gpu.py compares the two).
_NO_READBACK = True
import os as _os
_FIXED_TRAIN = _os.environ.get("MMT_FIXED_TRAIN_LISTS", "1") != "0"   # round 6 (A/B timing): the TRAINING lists at fixed capacity too


def is_teacher_fpn(backbone):
    """the FPN of a teacher model (built with out_planes False: its levels do not all feed an RPN head)"""
    fpn = getattr(backbone, "fpn", None)
    return fpn is not None and getattr(fpn, "out_planes", True) is not True


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg, is_teacher=False, is_student=False):
        super().__init__()
        self.cfg = cfg
        self.backbone = build_backbone(cfg)
        if is_teacher and hasattr(self.backbone, "fpn"):
            self.backbone.fpn.out_planes = False  # layers/fused.py::FPNFn
        self.rpn = build_rpn(cfg, is_teacher)
        self.box_heads = box_roi_heads(cfg)
        self.mask_heads = mask_roi_heads(cfg, is_student)
        # generalized_rcnn.py:25-30 (module order = parameter order = the order the EMA zips by)
        self.relation_nms = DuplicationRemovalNetwork(cfg, is_teacher) if cfg.MODEL.RELATION_NMS.USE_RELATION_NMS else None
        self.mt_fg_hint = cfg.MT.FG_HINT
        self.mt_cls = cfg.MT.CLS_LOSS
        self.hint_adaptor = MaskRCNNFPNAdaptor(cfg)
        self.taps = None  # dict: records stage outputs (tests / debugging)

    # ---- test instrumentation: replay of recorded random decisions (sampler index sets, dropout masks); see utils/replay.py
    def run_backbone(self, x, slot=0):
        """backbone(x) as a tuple (`slot`: which of the passes of one step, kept for the engine's call sites).  A no-grad pass on the
        default arithmetic goes through a launch plan (_hip.planned: recorded once per shape, replayed afterwards): the teacher's
        K x flip batch every step"""
        from maskrcnn_benchmark import _hip
        if (_hip.LAUNCH_PLANS and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                and _hip.PROFILE is None and _hip.F16X2 and _hip.get_conv_precision() == 3 and not _hip.bf16_storage()
                and getattr(self.backbone.body, "grad_ready", None) is None
                and self.backbone.body.stem.fused_ok(x, count=False)):
            # (only the fused stem reads the image through C-ABI launches alone; the key carries what else selects
            # launches or addresses: the stem switch and this model's weight epoch -- bumped by load_state_dict)
            from maskrcnn_benchmark.modeling.backbone.backbone import _STEM_FUSED
            return _hip.planned(("backbone", id(self), _STEM_FUSED[0], getattr(self, "_weights_epoch", 0)),
                                lambda t: tuple(self.backbone(t)), x)
        return tuple(self.backbone(x))

    def load_state_dict(self, *args, **kwargs):
        """nn.Module.load_state_dict; the folded-BN / space-to-depth tensors and packed planes a recorded launch plan points at may be
        re-made afterwards, so plans recorded for the old weights are never replayed again"""
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        return super().load_state_dict(*args, **kwargs)

    def set_replay(self, replay):
        self._replay = replay
        fa = (lambda tag: replay.take_all(tag)) if replay is not None else None
        fn = (lambda tag: replay.take_next(tag)) if replay is not None else None
        self.rpn.loss_evaluator.fg_bg_sampler.replay = fa
        self.box_heads.box.loss_evaluator.fg_bg_sampler.replay = fa
        self.box_heads.box.feature_extractor.replay = fn

    def set_rng(self, generator):
        """one torch.Generator for every random draw of this model (fg/bg samplers, box-head dropout); None = global"""
        self.rpn.loss_evaluator.fg_bg_sampler.generator = generator
        self.box_heads.box.loss_evaluator.fg_bg_sampler.generator = generator
        self.box_heads.box.feature_extractor.generator = generator

    def _tap(self, name, value):
        rp = getattr(self, "_replay", None)
        if rp is not None and rp.substitute_lists == "where_different" and rp.has(name):
            # reduced-precision runs against the fp32 oracle: the product's own list is kept wherever its discrete decisions
            # agree with the record (same count, same boxes to 1 px after the near-tie alignment) and replaced -- and
            # reported in taps[name + "_substituted"] -- only where they do not (the recorded sampler positions would then
            # refer to other boxes)
            aligned, moved = rp.align(name, value, tol=1.0)
            rec = rp.d[name]
            same = len(rec) == len(aligned) and all(
                len(b) == r[0].shape[0] and (len(b) == 0 or float((b.bbox - r[0].to(b.bbox.device)).abs().max()) <= 1.0)
                for b, r in zip(aligned, rec))
            if self.taps is not None:
                self.taps[name + "_substituted"] = not same
                # how far apart the two lists are as SETS: share of the recorded boxes that the product also produced (1 px)
                found = total = 0
                for b, r in zip(value, rec):
                    rb = r[0].to(b.bbox.device)
                    total += rb.shape[0]
                    if rb.shape[0] and len(b):
                        found += int(((rb[:, None, :] - b.bbox[None, :, :]).abs().amax(2) <= 1.0).any(1).sum())
                self.taps[name + "_agreement"] = found / max(total, 1)
            if same:
                value = aligned
                if self.taps is not None:
                    self.taps[name + "_moved"] = moved
                    self.taps[name] = value
                return value
        if rp is not None and rp.substitute_lists and rp.has(name):
            from maskrcnn_benchmark.structures.bounding_box import BoxList
            rec, new = rp.take_all(name), []
            for r, old in zip(rec, value):
                b = BoxList(r[0].to(old.bbox.device), old.size, "xyxy")
                if name == "detections":
                    b.add_field("scores", r[1].to(old.bbox.device))
                    b.add_field("objectness", r[3 if len(r) > 3 else 1].to(old.bbox.device))
                    b.add_field("labels", r[2].to(old.bbox.device))
                else:
                    b.add_field("objectness", r[1].to(old.bbox.device))
                new.append(b)
            value = new
        elif rp is not None and rp.has(name):
            value, moved = rp.align(name, value)  # order of near-tied candidates only; never values (utils/replay.py)
            if self.taps is not None:
                self.taps[name + "_moved"] = moved
        if self.taps is not None:
            self.taps[name] = value
        return value

    def _lists_free(self):
        """nobody outside the heads looks at the proposal / detection lists of this pass: they may stay at fixed capacity"""
        return (self.taps is None and getattr(self, "_replay", None) is None and self.relation_nms is None
                and not self.mask_heads.mask.use_relation)

    def set_module_mode(self, mode):
        self.rpn.set_teacher_mode(mode)
        self.box_heads.box.set_teacher_mode(mode)
        if self.relation_nms is not None:
            self.relation_nms.set_teacher_mode(mode)
        self.mask_heads.mask.set_teacher_mode(mode)

    def forward(self, images, targets=None, tta=None, features=None):
        """`features`: optional precomputed backbone(images.tensors) pyramid (the engine batches the backbone pass of
        the labeled and unlabeled student crops; the teacher reuses view 0's pyramid for its coarse inference)."""
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        if features is None:
            features = self.backbone(images.tensors)
        f_rpn = f_box = f_mask = features
        if self.training and torch.is_grad_enabled():
            # three consumers of every pyramid level: one launch sums their gradients (layers/fused.py::ForkFn)
            f_rpn, f_box, f_mask = fused.fork_levels(features, 3)
        # SURVEY f-2 (round 6): the training lists at fixed capacity, counts consumed on the device -- no read-back between the RPN
        # head and the box head's losses (rpn.py::select, box_head.py::subsample_fixed).  Tests that look at the lists (taps /
        # replay) and IR-Net (its modules take the sliced lists) keep the reference's sliced form
        sel = self.rpn.box_selector_train
        fixed_train = (self.training and _NO_READBACK and _FIXED_TRAIN and self._lists_free() and features[0].is_cuda
                       and not getattr(sel, "fixed_capacity", False))
        if fixed_train:
            sel.fixed_capacity = True
        try:
            proposals, proposal_losses = self.rpn(images, f_rpn, targets)
        finally:
            if fixed_train:
                sel.fixed_capacity = False
        proposals = self._tap("rpn_proposals" if self.training else "infer_proposals", proposals)
        x, result, losses, class_logits, box_regression = self.box_heads(f_box, proposals, targets)
        nms_loss = None
        if self.relation_nms is not None:
            result, nms_loss = self._relation_nms(x, result, class_logits, box_regression, targets)
        if not self.training:
            result = self._tap("detections", result)
        result, detector_losses = self.mask_heads(losses, f_mask, result, targets, images)
        if self.training:
            out = {}
            out.update(detector_losses)
            out.update(proposal_losses)
            if nms_loss is not None:
                out.update(nms_loss)
            return out
        return result

    def _relation_nms(self, x, result, class_logits, box_regression, targets):
        """generalized_rcnn.py:62-96: learned duplicate removal on the (un-NMS'ed) box-head output -- all images in one pass
        (modeling/relation/relation_module.py::forward_batch; the reference loops over images)"""
        prob = torch.softmax(class_logits, dim=1)
        # `x` is the fused ReLU(+dropout) output of fc7: its consumers mask the gradient (layers/fused.py convention)
        self.relation_nms.in_mask_scale = self.box_heads.box._scale(self.training)
        outs, loss = self.relation_nms.forward_batch(x, result, prob, box_regression, targets)
        if self.training:
            return result, loss
        return outs, None

    def forward_teacher(self, images, targets=None):
        if targets is not None:
            # reference generalized_rcnn.py:133-139: ground truth in place of the coarse inference, its masks decoded at a hard-coded
            # 800 x 800 (`get_field('masks').decode(800, 800)`); engine/MTtrainer.py:247-275 never passes targets
            raise NotImplementedError("forward_teacher(images, targets): the ground-truth branch of the reference "
                                      "(generalized_rcnn.py:133-139) is not built; MTtrainer never takes it")
        integral = []
        images = [to_image_list(im) for im in images]
        # all AUG_K x {plain, mirrored} views go through the backbone as ONE batch; the coarse inference of the
        # reference (a second, identical backbone pass on view 0, generalized_rcnn.py:126-127) reuses pyramid 0
        first = ImageListView(images[0])
        aug_features = self.extract_aug_feat(images)
        cb = self.__dict__.get("on_backbone_issued")   # the training engine's cue: from here on the teacher issues small kernels
        if cb is not None:
            cb()
        batched = getattr(self, "_batched_pyr", None)
        # RPN head outputs and decoded+NMS'ed candidates are shared between the two selector passes on pyramid 0
        self.rpn.shared = {"pre": max(self.rpn.box_selector_train.pre_nms_top_n, self.rpn.box_selector_test.pre_nms_top_n)}
        self.set_module_mode("test")
        # the coarse inference's proposal lists at fixed capacity, no count read-back (rpn.py::select); with taps / replay (tests
        # that look at these lists) and with IR-Net (its modules take the lists) the lists are sliced as in the reference
        sel, post = self.rpn.box_selector_test, self.box_heads.box.post_processor
        fixed = (_NO_READBACK and self.taps is None and getattr(self, "_replay", None) is None and self.relation_nms is None
                 and not self.mask_heads.mask.use_relation and self.mask_heads.mask.mask_generator.masker is not None
                 and aug_features[0][0].is_cuda)
        sel.fixed_capacity = post.defer_counts = fixed
        try:
            teacher_infer = self.forward(first, features=aug_features[0])
        finally:
            sel.fixed_capacity = post.defer_counts = False
        if self.mt_fg_hint > 0:
            for t in teacher_infer:
                integral.append(t.get_field("mask").sum(0)[0])
            for t in teacher_infer:
                t.remove_field("mask")
        if fixed:
            # the detection counts, read back with the mask head and the paste already queued behind the box head
            from ..roi_heads.box_head.box_head import resolve_counts
            teacher_infer = resolve_counts(teacher_infer)
        self.set_module_mode("train")
        sel_t = self.rpn.box_selector_train
        sel_t.fixed_capacity = fixed and _FIXED_TRAIN   # (round 6: the train-config lists too -- the box head's sampler takes the counts on the device)
        try:
            _, _, _, _, proposals, _, ffi_boxes = self.rpn.forward_teacher(images[0], aug_features[0], teacher_infer)
        finally:
            sel_t.fixed_capacity = False
        self.rpn.shared = None
        proposals = self._tap("teacher_proposals", proposals)
        embeddings = self.get_emb_feature(aug_features) if self.cfg.MT.FG_HINT else None
        result, class_logits = None, None
        if self.cfg.MT.CLS_LOSS:
            self.box_heads.box.batched_pyramid = batched
            _, result, _, class_logits, _ = self.box_heads.forward_teacher(aug_features, proposals, teacher_infer)
            self.box_heads.box.batched_pyramid = None
        return {"result_t": result, "class_logit_t": class_logits, "embedding": embeddings, "seg_mask": integral,
                "ffi_boxes": ffi_boxes}

    def forward_student(self, images, result_t, features=None, embeddings=None):
        """`embeddings`: the student's hint-adaptor outputs when the caller already computed them (they do not depend on
        the teacher, so the trainer launches them before it waits for the teacher)"""
        images = [to_image_list(im) for im in images] if isinstance(images, list) else [to_image_list(images)]
        feat_list = features if features is not None else self.extract_aug_feat(images, teacher=False)
        loss_dict = {}
        if self.cfg.MT.FG_HINT:
            emb = embeddings if embeddings is not None else self.get_emb_feature(feat_list)
            loss_dict.update(mt_fg_loss=fg_hint_loss(result_t["embedding"], emb, result_t["seg_mask"]))
        if self.cfg.MT.CLS_LOSS:
            loss_dict.update(self.box_heads.forward_student(feat_list, result_t["result_t"], result_t["class_logit_t"]))
        return loss_dict

    def extract_aug_feat(self, imglist, teacher=True):
        """generalized_rcnn.py:201-215 (ImageList.hflip mutates in place, as in the reference)"""
        feats = []
        if teacher:
            views = []
            for img in imglist:
                views.append(img.tensors)
                img.hflip()
                views.append(img.tensors)
            n = views[0].shape[0]
            if all(v.shape == views[0].shape for v in views):
                if is_teacher_fpn(self.backbone):
                    # only view 0's pyramid feeds a plane-fed launch (the coarse inference's RPN head): the FPN's output convolutions
                    # write row-blocked planes for the first n images of the K x flip batch (layers/fused.py::fpn_forward)
                    self.backbone.fpn.out_planes = n
                pyr = self.run_backbone(torch.cat(views, 0))
                self._batched_pyr = (pyr, n, len(views))
                # (batch_slice: the statistics slot / planes of a level go along with its slices -- no reduction pass per level
                # in front of the RPN head of the coarse inference)
                feats = [tuple(fused.batch_slice(level, i * n, (i + 1) * n) for level in pyr) for i in range(len(views))]
            else:
                feats = [self.backbone(v) for v in views]
        else:
            for i, img in enumerate(imglist):
                if i % 2 == 1:
                    img.hflip()
                feats.append(self.backbone(img.tensors))
        return feats

    def get_emb_feature(self, feature_list):
        bp = getattr(self, "_batched_pyr", None)
        if bp is not None and len(feature_list) == bp[2] and feature_list[0][0].data_ptr() == bp[0][0].data_ptr():
            # the views came out of one batched backbone pass: run the 5 adaptor convs once on the batched levels
            pyr, n, nv = bp
            self._batched_pyr = None
            emb = self.hint_adaptor(pyr)
            return [[e[i * n:(i + 1) * n] for e in emb] for i in range(nv)]
        return [self.hint_adaptor(f) for f in feature_list]

    def get_fg_feature_loss(self, feature_list, seg_mask, teacher_feat):
        return fg_hint_loss(teacher_feat, self.get_emb_feature(feature_list), seg_mask)


def fg_hint_loss(teachers, students, masks):
    """MGD (generalized_rcnn.py:243-282): teachers = list over views of 5-level embeddings (odd views were
    computed on mirrored inputs and are un-mirrored inside the kernel), students = [5-level embeddings]."""
    if len(students) != 1:
        raise NotImplementedError("MT.AUG_S > 1: the mirrored student views of the reference (generalized_rcnn.py:253-255, 274-280; "
                                  "forward_student's per-view loop, :164-205) are not built -- every shipped recipe has AUG_S = 1")
    seg = torch.stack([m.to(torch.int32) for m in masks]).contiguous()
    s = students[0]
    nl = len(s)
    flips = [i % 2 == 1 for i in range(len(teachers))]
    flat_t = [t[l].detach() for t in teachers for l in range(nl)]
    return fused.MGDLossFn.apply(seg, flips, nl, *s, *flat_t)
