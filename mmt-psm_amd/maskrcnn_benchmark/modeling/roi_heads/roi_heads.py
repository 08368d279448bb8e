"""BoxROIHeads / MaskROIHeads glue (reference: modeling/roi_heads/roi_heads.py:42-99)."""
import torch

from .box_head.box_head import build_roi_box_head
from .mask_head.mask_head import build_roi_mask_head


class BoxROIHeads(torch.nn.ModuleDict):
    def __init__(self, cfg, heads):
        super().__init__(heads)
        self.cfg = cfg.clone()

    def forward(self, features, proposals, targets=None):
        x, detections, loss_box, class_logits, box_regression = self.box(features, proposals, targets)
        return x, detections, dict(loss_box), class_logits, box_regression

    def forward_student(self, features, proposals, class_logits_t):
        return self.box.forward_student(features, proposals, class_logits_t)

    def forward_teacher(self, feature_tuple, proposals, teacher_infer):
        x, detections, loss_box, class_logits, box_regression = self.box.forward_teacher(
            feature_tuple, proposals, targets=teacher_infer)
        return x, detections, dict(loss_box), class_logits, box_regression


class MaskROIHeads(torch.nn.ModuleDict):
    def __init__(self, cfg, heads):
        super().__init__(heads)
        self.cfg = cfg

    def forward(self, losses, features, detections, targets=None, images=None):
        _, detections, loss_mask = self.mask(features, detections, targets, images)
        losses.update(loss_mask)
        return detections, losses


def box_roi_heads(cfg, relation=True):
    return BoxROIHeads(cfg, [("box", build_roi_box_head(cfg, relation=relation))])


def mask_roi_heads(cfg, is_student=False):
    return MaskROIHeads(cfg, [("mask", build_roi_mask_head(cfg, is_student))])
