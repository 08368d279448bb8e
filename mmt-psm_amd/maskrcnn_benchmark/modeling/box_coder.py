"""BoxCoder (reference: modeling/box_coder.py:7-95): (dx,dy,dw,dh) <-> xyxy with the +1 pixel convention."""
import math

import torch


class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode(self, reference_boxes, proposals):
        pw = proposals[:, 2] - proposals[:, 0] + 1
        ph = proposals[:, 3] - proposals[:, 1] + 1
        px = proposals[:, 0] + 0.5 * pw
        py = proposals[:, 1] + 0.5 * ph
        gw = reference_boxes[:, 2] - reference_boxes[:, 0] + 1
        gh = reference_boxes[:, 3] - reference_boxes[:, 1] + 1
        gx = reference_boxes[:, 0] + 0.5 * gw
        gy = reference_boxes[:, 1] + 0.5 * gh
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (gx - px) / pw, wy * (gy - py) / ph, ww * torch.log(gw / pw),
                            wh * torch.log(gh / ph)), dim=1)

    def decode(self, rel_codes, boxes):
        if (rel_codes.is_cuda and rel_codes.dtype == torch.float32 and not rel_codes.requires_grad and rel_codes.dim() == 2
                and rel_codes.shape[1] % 4 == 0 and rel_codes.shape[0] > 0):
            from maskrcnn_benchmark import _hip as H   # one launch (mmt_box_decode) instead of ~30 elementwise ones
            return H.box_decode(rel_codes, boxes, self.weights, self.bbox_xform_clip)
        boxes = boxes.to(rel_codes.dtype)
        w = (boxes[:, 2] - boxes[:, 0] + 1)[:, None]
        h = (boxes[:, 3] - boxes[:, 1] + 1)[:, None]
        cx = boxes[:, 0, None] + 0.5 * w
        cy = boxes[:, 1, None] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx, pcy = dx * w + cx, dy * h + cy
        pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
        out = torch.zeros_like(rel_codes)
        out[:, 0::4] = pcx - 0.5 * pw
        out[:, 1::4] = pcy - 0.5 * ph
        out[:, 2::4] = pcx + 0.5 * pw - 1
        out[:, 3::4] = pcy + 0.5 * ph - 1
        return out
