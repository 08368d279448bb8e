"""BoxList operations (reference: structures/boxlist_ops.py:9-134)."""
import torch

from .bounding_box import BoxList


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="score"):
    """Greedy NMS on one BoxList through `_C.nms` (boxlist_ops.py:9-35).  The batched proposal pipeline uses
    `_hip.nms_batched` directly; this wrapper exists for API parity and for callers outside it."""
    if nms_thresh <= 0:
        return boxlist
    from maskrcnn_benchmark.layers import nms as _box_nms
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)


def remove_small_boxes(boxlist, min_size):
    wh = boxlist.convert("xywh").bbox
    keep = ((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).nonzero().squeeze(1)
    return boxlist[keep]


def box_iou_tensor(b1, a1, b2, a2):
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def boxlist_iou(boxlist1, boxlist2):
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    return box_iou_tensor(boxlist1.bbox, boxlist1.area(), boxlist2.bbox, boxlist2.area())


def cat_boxlist(bboxes):
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size, mode = bboxes[0].size, bboxes[0].mode
    assert all(b.size == size and b.mode == mode for b in bboxes)
    fields = set(bboxes[0].fields())
    assert all(set(b.fields()) == fields for b in bboxes)
    out = BoxList(torch.cat([b.bbox for b in bboxes], 0) if len(bboxes) > 1 else bboxes[0].bbox, size, mode)
    for f in bboxes[0].fields():
        if f == "mask":
            continue
        out.add_field(f, torch.cat([b.get_field(f) for b in bboxes], 0) if len(bboxes) > 1 else bboxes[0].get_field(f))
    return out
