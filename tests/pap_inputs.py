"""Synthetic instance masks for the PAP evaluator tests: 7 windows of 96 x 96, two categories; detections are perturbed copies
of the ground truths plus false positives, with misses, duplicated claims (two ground truths whose best detection is the
same), an exact-duplicate detection (equal IoUs), a window with ground truth but no detection of a category, one with
detections but no ground truth, and a window whose numeric id is 0 (the evaluator's `> 0` match marker).  Dense uint8 masks;
the caller encodes them with the RLE codec under test (tests/golden/gen_golden.py: the reference's pycocotools)."""
import numpy as np

SIZE = 96


def _ellipse(cx, cy, rx, ry):
    yy, xx = np.mgrid[0:SIZE, 0:SIZE]
    return ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0).astype(np.uint8)


def make(seed=7):
    """-> (gts, dts): lists of dicts with a dense 'mask' (to be encoded as 'segmentation')"""
    rng = np.random.RandomState(seed)
    gts, dts = [], []
    for w in range(7):
        image_id = {"file_name": "slide%d" % (w % 3), "location": (128 * w, 64 * (w % 2)), "id": w}
        for cat in (1, 2):
            n_gt = 0 if (w == 5 and cat == 1) else int(rng.randint(4, 9))
            these = []
            for _ in range(n_gt):
                r = rng.uniform(5, 11) if cat == 1 else rng.uniform(3, 6)
                m = _ellipse(rng.uniform(12, SIZE - 12), rng.uniform(12, SIZE - 12), r, r * rng.uniform(0.7, 1.3))
                these.append(m)
                gts.append({"image_id": image_id, "category_id": cat, "mask": m})
            if w == 4 and cat == 2:
                continue                                   # ground truth, no detections
            for m in these:
                u = rng.rand()
                if u < 0.15:
                    continue                               # missed
                ys, xs = np.nonzero(m)
                cx, cy = xs.mean() + rng.uniform(-2.5, 2.5), ys.mean() + rng.uniform(-2.5, 2.5)
                rx = (xs.max() - xs.min() + 1) / 2.0 * rng.uniform(0.8, 1.25)
                ry = (ys.max() - ys.min() + 1) / 2.0 * rng.uniform(0.8, 1.25)
                dts.append({"image_id": image_id, "category_id": cat, "mask": _ellipse(cx, cy, rx, ry), "score": float(rng.rand())})
            if these and w in (1, 3):                      # one detection covering two neighbouring ground truths
                a = these[0]
                b = np.roll(a, 5, axis=1)
                gts.append({"image_id": image_id, "category_id": cat, "mask": b})
                dts.append({"image_id": image_id, "category_id": cat, "mask": (a | b).astype(np.uint8), "score": float(rng.rand())})
            if these and w == 2:                           # an exact duplicate of a detection (equal IoUs everywhere)
                d = dict(dts[-1])
                d["score"] = float(rng.rand())
                dts.append(d)
            for _ in range(int(rng.randint(0, 3))):        # false positives
                dts.append({"image_id": image_id, "category_id": cat, "score": float(rng.rand()),
                            "mask": _ellipse(rng.uniform(8, SIZE - 8), rng.uniform(8, SIZE - 8), rng.uniform(2, 6), rng.uniform(2, 6))})
    return gts, dts
