"""LevelMapper / Pooler (reference: modeling/poolers.py:11-121).  The per-level loop of the reference
(nonzero -> gather -> ROIAlign -> index_put, x4) is one fused launch here (`fused.RoiAlignFpnFn`)."""
import math

import torch
from torch import nn

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import fused


class LevelMapper(object):
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([b.area() for b in boxlists]))
        lv = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        return torch.clamp(lv, min=self.k_min, max=self.k_max).to(torch.int64) - self.k_min


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = sampling_ratio
        self.map_levels = LevelMapper(-math.log2(self.scales[0]), -math.log2(self.scales[-1]))

    @staticmethod
    def convert_to_roi_format(boxes):
        bb = torch.cat([b.bbox for b in boxes], 0)
        ids = torch.cat([torch.full((len(b), 1), i, dtype=bb.dtype, device=bb.device) for i, b in enumerate(boxes)], 0)
        return torch.cat([ids, bb], 1)

    def forward(self, x, boxes):
        """x: per-level feature maps (N,C,H,W); boxes: list[BoxList] -> (R, C, res, res) NHWC-dense"""
        n_lv = len(self.scales)
        m = self.map_levels
        if all(b.mode == "xyxy" for b in boxes) and len(boxes) <= 32 and boxes[0].bbox.is_cuda:
            # roi format and level of every box of the call in ONE launch (same fp32 expressions as the tensor code below)
            rois, levels = H.roi_format_levels([b.bbox for b in boxes], m.s0, m.lvl0, m.eps, m.k_min, m.k_max)
            if n_lv == 1:
                levels.zero_()
        else:
            rois = self.convert_to_roi_format(boxes)
            levels = self.map_levels(boxes).to(torch.int32) if n_lv > 1 else torch.zeros(
                (rois.shape[0],), dtype=torch.int32, device=rois.device)
        return fused.RoiAlignFpnFn.apply(rois, levels, self.output_size[0], self.scales, self.sampling_ratio,
                                         *list(x)[:n_lv])
