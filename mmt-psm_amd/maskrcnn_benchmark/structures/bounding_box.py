"""BoxList: the container that crosses every boundary of the path (reference:
structures/bounding_box.py:9-266).  Same constructor, fields API, `+1` pixel convention, `size=(W,H)`.
Re-implemented around one (n,4) fp32 tensor that stays on the device; nothing here calls the host."""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1
_TO_REMOVE = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        dev = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=dev)
        if bbox.ndimension() != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be (n,4), got {}".format(tuple(bbox.shape)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox = bbox
        self.size = image_size
        self.mode = mode
        self.extra_fields = {}

    # ---- fields
    def add_field(self, name, data):
        self.extra_fields[name] = data

    def get_field(self, name):
        return self.extra_fields[name]

    def has_field(self, name):
        return name in self.extra_fields

    def remove_field(self, name):
        del self.extra_fields[name]

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        self.extra_fields.update(other.extra_fields)

    def copy_with_fields(self, fields):
        out = BoxList(self.bbox, self.size, self.mode)
        for f in ([fields] if not isinstance(fields, (list, tuple)) else fields):
            out.add_field(f, self.get_field(f))
        return out

    # ---- geometry
    def _xyxy(self):
        if self.mode == "xyxy":
            return self.bbox.unbind(-1)
        x, y, w, h = self.bbox.unbind(-1)
        return x, y, x + (w - _TO_REMOVE).clamp(min=0), y + (h - _TO_REMOVE).clamp(min=0)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._xyxy()
        if mode == "xyxy":
            b = torch.stack((x1, y1, x2, y2), -1)
        else:
            b = torch.stack((x1, y1, x2 - x1 + _TO_REMOVE, y2 - y1 + _TO_REMOVE), -1)
        out = BoxList(b, self.size, mode)
        out._copy_extra_fields(self)
        return out

    def _map_fields(self, out, fn):
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else fn(v))
        return out

    def resize(self, size, *a, **kw):
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        x1, y1, x2, y2 = self._xyxy()
        if rw == rh:
            out = BoxList(self.bbox * rw, size, self.mode)
            return self._map_fields(out, lambda v: v.resize(size, *a, **kw))
        out = BoxList(torch.stack((x1 * rw, y1 * rh, x2 * rw, y2 * rh), -1), size, "xyxy")
        return self._map_fields(out, lambda v: v.resize(size, *a, **kw)).convert(self.mode)

    def transpose(self, method):
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        x1, y1, x2, y2 = self._xyxy()
        if method == FLIP_LEFT_RIGHT:
            b = torch.stack((w - x2 - _TO_REMOVE, y1, w - x1 - _TO_REMOVE, y2), -1)
        else:
            b = torch.stack((x1, h - y2, x2, h - y1), -1)
        out = BoxList(b, self.size, "xyxy")
        return self._map_fields(out, lambda v: v.transpose(method)).convert(self.mode)

    def crop(self, box):
        x1, y1, x2, y2 = self._xyxy()
        w, h = box[2] - box[0], box[3] - box[1]
        b = torch.stack(((x1 - box[0]).clamp(min=0, max=w), (y1 - box[1]).clamp(min=0, max=h),
                         (x2 - box[0]).clamp(min=0, max=w), (y2 - box[1]).clamp(min=0, max=h)), -1)
        out = BoxList(b, (w, h), "xyxy")
        return self._map_fields(out, lambda v: v.crop(box)).convert(self.mode)

    def clip_to_image(self, remove_empty=True):
        w, h = self.size
        if self.bbox.is_cuda and self.bbox.dtype == torch.float32 and self.bbox.dim() == 2:
            # one launch instead of four column clamps: the same comparisons against per-column bounds, which travel through
            # the pinned ring (no blocking pageable copy for an image size not seen before, nothing cached per size: ADVICE r3)
            from maskrcnn_benchmark.utils.miscellaneous import dev_const, dev_floats
            lo = dev_const([0.0, 0.0, 0.0, 0.0], torch.float32, self.bbox.device)
            hi = dev_floats([w - _TO_REMOVE, h - _TO_REMOVE, w - _TO_REMOVE, h - _TO_REMOVE], self.bbox.device)
            torch.clamp(self.bbox, min=lo, max=hi, out=self.bbox)
        else:
            self.bbox[:, 0].clamp_(min=0, max=w - _TO_REMOVE)
            self.bbox[:, 1].clamp_(min=0, max=h - _TO_REMOVE)
            self.bbox[:, 2].clamp_(min=0, max=w - _TO_REMOVE)
            self.bbox[:, 3].clamp_(min=0, max=h - _TO_REMOVE)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + _TO_REMOVE) * (b[:, 3] - b[:, 1] + _TO_REMOVE)
        return b[:, 2] * b[:, 3]

    # ---- tensor-like
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            len(self), self.size[0], self.size[1], self.mode)
