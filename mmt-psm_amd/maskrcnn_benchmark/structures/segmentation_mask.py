"""Polygon instance masks (reference: structures/segmentation_mask.py:53-247).

The reference keeps polygons as Python lists of small CPU tensors and rasterises them one ROI at a time
through pycocotools (mask_head/loss.py:37-75).  Here a SegmentationMask also owns a packed device copy
(vertex array + offsets) so that crop / resize / rasterise of all positive ROIs is ONE kernel launch
(`_hip.polygon_targets`); the list-of-tensors view is kept for API parity (indexing, iteration, flips).
"""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class Polygons(object):
    def __init__(self, polygons, size, mode=None):
        if isinstance(polygons, Polygons):
            polygons = polygons.polygons
        self.polygons = [torch.as_tensor(p, dtype=torch.float32).reshape(-1) for p in polygons]
        self.size = size
        self.mode = mode

    def transpose(self, method):
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        dim, idx = (w, 0) if method == FLIP_LEFT_RIGHT else (h, 1)
        out = []
        for p in self.polygons:
            q = p.clone()
            q[idx::2] = dim - p[idx::2] - 1
            out.append(q)
        return Polygons(out, self.size, self.mode)

    def crop(self, box):
        w, h = max(box[2] - box[0], 1), max(box[3] - box[1], 1)
        out = []
        for p in self.polygons:
            q = p.clone()
            q[0::2] = q[0::2] - box[0]
            q[1::2] = q[1::2] - box[1]
            out.append(q)
        return Polygons(out, (w, h), self.mode)

    def resize(self, size, *a, **kw):
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        out = []
        for p in self.polygons:
            q = p.clone()
            q[0::2] *= rw
            q[1::2] *= rh
            out.append(q)
        return Polygons(out, size, self.mode)

    def __repr__(self):
        return "Polygons(num_polygons={}, image_width={}, image_height={}, mode={})".format(
            len(self.polygons), self.size[0], self.size[1], self.mode)


class SegmentationMask(object):
    def __init__(self, polygons, size, mode=None):
        assert isinstance(polygons, list)
        self.polygons = [p if isinstance(p, Polygons) else Polygons(p, size, mode) for p in polygons]
        self.size = size
        self.mode = mode
        self._packed = None

    def packed(self, device):
        """-> (xy float32 [V,2] flattened, poly_off int32 [NP+1], inst_range int32 [G,2]) on `device`"""
        if self._packed is None or self._packed[0].device != torch.device(device):
            xy, off, rng = [], [0], []
            for inst in self.polygons:
                b = len(off) - 1
                for p in inst.polygons:
                    xy.append(p)
                    off.append(off[-1] + p.numel() // 2)
                rng.append([b, len(off) - 1])
            xy = torch.cat(xy) if xy else torch.zeros(0)
            self._packed = (xy.to(device), torch.tensor(off, dtype=torch.int32, device=device),
                            torch.tensor(rng, dtype=torch.int32, device=device).reshape(-1, 2))
        return self._packed

    def transpose(self, method):
        return SegmentationMask([p.transpose(method) for p in self.polygons], self.size, self.mode)

    def crop(self, box):
        w, h = box[2] - box[0], box[3] - box[1]
        return SegmentationMask([p.crop(box) for p in self.polygons], (w, h), self.mode)

    def resize(self, size, *a, **kw):
        return SegmentationMask([p.resize(size, *a, **kw) for p in self.polygons], size, self.mode)

    def to(self, *a, **kw):
        return self

    def __getitem__(self, item):
        if isinstance(item, (int, slice)):
            sel = self.polygons[item]
            sel = sel if isinstance(sel, list) else [sel]
        else:
            if isinstance(item, torch.Tensor) and item.dtype in (torch.uint8, torch.bool):
                item = item.nonzero().reshape(-1)
            sel = [self.polygons[int(i)] for i in (item.tolist() if isinstance(item, torch.Tensor) else item)]
        return SegmentationMask(sel, self.size, self.mode)

    def __iter__(self):
        return iter(self.polygons)

    def __len__(self):
        return len(self.polygons)

    def __repr__(self):
        return "SegmentationMask(num_instances={}, image_width={}, image_height={})".format(
            len(self.polygons), self.size[0], self.size[1])
