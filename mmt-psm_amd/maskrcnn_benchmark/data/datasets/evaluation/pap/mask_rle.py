"""Run-length masks for the PAP evaluator: the five calls the reference makes into its vendored pycocotools
(`/root/reference/pycoco/_mask.pyx`: encode :144, decode :160, merge :177, area :190, iouIntUni :293-380; C side
`maskApi.c`: rleEncode :21-34, rleToString / rleFrString :204-236, rleIouInterUnion :239-260) restated on numpy.

An RLE is COCO's: {"size": [h, w], "counts": ...} over the mask flattened COLUMN-major, runs alternating 0 / 1 starting with
zeros; `counts` is the compressed ASCII string (bytes or str) or a plain list of run lengths (COCO's "uncompressed" form).
Host code: evaluation runs once per checkpoint on a few hundred windows ("irrelevant to the throughput
metric")."""
import numpy as np


def _runs(flat):
    """flat uint8 {0,1} -> run lengths, first run = zeros (possibly of length 0)"""
    n = flat.shape[0]
    if n == 0:
        return [0]
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate(([0], change, [n]))
    runs = np.diff(edges).tolist()
    if flat[0]:
        runs = [0] + runs
    return runs


def _to_string(cnts):
    """maskApi.c:204-217: LEB128-like, 5 data bits + continuation bit per char (ASCII 48..111); from the third run on the
    difference to the run two places back is stored"""
    out = []
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def _from_string(s):
    """maskApi.c:219-236"""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    cnts, p, n = [], 0, len(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def counts_of(rle):
    c = rle["counts"]
    return [int(v) for v in c] if isinstance(c, (list, tuple, np.ndarray)) else _from_string(c)


def encode(mask):
    """(h, w) -> RLE; (h, w, n) -> list of RLEs (`_mask.pyx:144`); counts as bytes like pycocotools"""
    m = np.asarray(mask)
    if m.ndim == 3:
        return [encode(m[:, :, i]) for i in range(m.shape[2])]
    h, w = m.shape
    runs = _runs((m != 0).astype(np.uint8).flatten(order="F"))
    return {"size": [int(h), int(w)], "counts": _to_string(runs).encode("ascii")}


def decode(rle):
    """RLE -> (h, w) uint8 in Fortran order; list of RLEs -> (h, w, n)"""
    if isinstance(rle, (list, tuple)):
        return np.stack([decode(r) for r in rle], axis=2) if len(rle) else np.zeros((0, 0, 0), np.uint8)
    h, w = rle["size"]
    cnts = counts_of(rle)
    vals = np.zeros(len(cnts), dtype=np.uint8)
    vals[1::2] = 1
    flat = np.repeat(vals, cnts)
    if flat.shape[0] != h * w:
        raise ValueError("RLE does not cover its %d x %d mask" % (h, w))
    return flat.reshape((h, w), order="F")


def area(rle):
    if isinstance(rle, (list, tuple)):
        return np.array([area(r) for r in rle], dtype=np.uint32)
    return np.uint32(sum(counts_of(rle)[1::2]))


def merge(rles, intersect=False):
    """union (or intersection) of several masks as one RLE (`_mask.pyx:177`)"""
    if not len(rles):
        raise ValueError("merge of an empty list")
    acc = decode(rles[0]).astype(bool)
    for r in rles[1:]:
        acc = (acc & decode(r).astype(bool)) if intersect else (acc | decode(r).astype(bool))
    return encode(acc.astype(np.uint8))


def _bbox(m):
    """rleToBbox (maskApi.c:135-151): [x, y, w, h] of the mask's extent; an empty mask is [0, 0, 0, 0].  The C code works on
    the runs: a run of ones that continues from the bottom of one column into the top of the next makes the box full height."""
    ys, xs = np.nonzero(m)
    if ys.size == 0:
        return 0.0, 0.0, 0.0, 0.0
    y0, y1 = int(ys.min()), int(ys.max())
    if m.shape[1] > 1 and bool(np.any(m[-1, :-1] & m[0, 1:])):
        y0, y1 = 0, m.shape[0] - 1
    return float(xs.min()), float(y0), float(xs.max() - xs.min() + 1), float(y1 - y0 + 1)


def iouIntUni(dt, gt, iscrowd):
    """(iou, intersection, union), each (len(dt), len(gt)) float64 -- the fork's addition to pycocotools (`_mask.pyx:293-380`,
    `maskApi.c:239-260`): pairs whose bounding boxes overlap get i = |d & g| and u = |d | g| (crowd gt: u = |d|) with the C
    code's `i == 0 -> u = 1`; a pair of different mask sizes gets iou -1.  Pairs whose boxes do not overlap have iou 0 and
    -- where the C code leaves its malloc'ed intersection / union cells unwritten -- intersection 0, union 0 here.
    [] when either list is empty."""
    m, n = len(dt), len(gt)
    if m == 0 or n == 0:
        return []
    D = [decode(r).astype(bool) for r in dt]
    G = [decode(r).astype(bool) for r in gt]
    bd, bg = [_bbox(x) for x in D], [_bbox(x) for x in G]
    iou, inter, uni = np.zeros((m, n)), np.zeros((m, n)), np.zeros((m, n))
    for g in range(n):
        for d in range(m):
            w = min(bd[d][0] + bd[d][2], bg[g][0] + bg[g][2]) - max(bd[d][0], bg[g][0])
            h = min(bd[d][1] + bd[d][3], bg[g][1] + bg[g][3]) - max(bd[d][1], bg[g][1])
            if w <= 0 or h <= 0:
                continue
            if D[d].shape != G[g].shape:
                iou[d, g] = -1
                continue
            i = int(np.count_nonzero(D[d] & G[g]))
            u = int(np.count_nonzero(D[d] | G[g]))
            if i == 0:
                u = 1
            elif iscrowd is not None and iscrowd[g]:
                u = int(np.count_nonzero(D[d]))
            iou[d, g], inter[d, g], uni[d, g] = i / u, float(i), float(u)
    return iou, inter, uni
