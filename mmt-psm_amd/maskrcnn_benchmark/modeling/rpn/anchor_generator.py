"""Anchors (reference: modeling/rpn/anchor_generator.py:34-265).  Cell anchors are computed in float64 numpy
exactly like the reference (round-half-even of np.round matters) and kept as buffers under the same
state-dict names; grid anchors + visibility are cached per (feature shape, image size, device)."""
import numpy as np
import torch
from torch import nn

from maskrcnn_benchmark.structures.bounding_box import BoxList


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    x0, y0, x1, y1 = 0.0, 0.0, stride - 1.0, stride - 1.0
    w, h = x1 - x0 + 1, y1 - y0 + 1
    cx, cy = x0 + 0.5 * (w - 1), y0 + 0.5 * (h - 1)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    rows = []
    for rw, rh in zip(ws, hs):  # ratio-major, then scale (anchor_generator.py:216-221)
        for s in scales:
            sw, sh = rw * s, rh * s
            rows.append([cx - 0.5 * (sw - 1), cy - 0.5 * (sh - 1), cx + 0.5 * (sw - 1), cy + 0.5 * (sh - 1)])
    return torch.from_numpy(np.array(rows, dtype=np.float64))


class BufferList(nn.Module):
    def __init__(self, buffers=None):
        super().__init__()
        for i, b in enumerate(buffers or []):
            self.register_buffer(str(i), b)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32),
                 straddle_thresh=0):
        super().__init__()
        if len(anchor_strides) == 1:
            cells = [generate_anchors(anchor_strides[0], sizes, aspect_ratios).float()]
        else:
            if len(anchor_strides) != len(sizes):
                raise RuntimeError("FPN should have #anchor_strides == #sizes")
            cells = [generate_anchors(st, (sz,), aspect_ratios).float() for st, sz in zip(anchor_strides, sizes)]
        self.strides = anchor_strides
        self.cell_anchors = BufferList(cells)
        self.straddle_thresh = straddle_thresh
        self._cache = {}

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def grid_anchors(self, grid_sizes):
        out = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            key = (int(gh), int(gw), stride, str(base.device), base._version)
            if key not in self._cache:
                sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=base.device)
                sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=base.device)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                sh = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
                self._cache[key] = (sh.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)
            out.append(self._cache[key])
        return out

    def visibility(self, anchors, image_width, image_height):
        t = self.straddle_thresh
        if t < 0:
            return torch.ones(anchors.shape[0], dtype=torch.bool, device=anchors.device)
        return ((anchors[:, 0] >= -t) & (anchors[:, 1] >= -t) & (anchors[:, 2] < image_width + t)
                & (anchors[:, 3] < image_height + t))

    def forward(self, image_list, feature_maps):
        """-> list (per image) of list (per level) of BoxList with field 'visibility' (anchor_generator.py:110-123).
        Anchors AND their visibility masks are constants of (grid sizes, image size): built once, then served from the
        cache (the fixed 1000x1000 crops of the path hit it every time; ~200 small launches per call otherwise)."""
        grids = tuple((int(f.shape[-2]), int(f.shape[-1])) for f in feature_maps)
        per_level = self.grid_anchors(grids)
        out = []
        for (ih, iw) in image_list.image_sizes:
            key = ("vis", grids, int(iw), int(ih), str(per_level[0].device), tuple(a.data_ptr() for a in per_level))
            vis = self._cache.get(key)
            if vis is None:
                vis = self._cache[key] = [self.visibility(a, iw, ih) for a in per_level]
            lv = []
            for a, v in zip(per_level, vis):
                b = BoxList(a, (iw, ih), mode="xyxy")
                b.add_field("visibility", v)
                lv.append(b)
            out.append(lv)
        return out


def make_anchor_generator(config):
    r = config.MODEL.RPN
    if r.USE_FPN:
        assert len(r.ANCHOR_STRIDE) == len(r.ANCHOR_SIZES), "FPN should have len(ANCHOR_STRIDE) == len(ANCHOR_SIZES)"
    else:
        assert len(r.ANCHOR_STRIDE) == 1, "Non-FPN should have a single ANCHOR_STRIDE"
    return AnchorGenerator(r.ANCHOR_SIZES, r.ASPECT_RATIOS, r.ANCHOR_STRIDE, r.STRADDLE_THRESH)
