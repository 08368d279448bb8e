"""Synthetic inputs of SURVEY.md section 8(d): COCO-style crops with polygon instances.

Stand-alone (torch + numpy only, no package imports) so that the benchmark, the tests and the
golden-vector generator all draw exactly the same data.  Mirrors the *output contract* of the
reference's data pipeline (data/collate_batch.py:5-77): labeled = images (N,3,H,W) fp32
BGR-255 mean-subtracted range + per image boxes/labels/polygons; unlabeled = AUG_K+AUG_S
colour-jittered copies of each base image.
"""
import math

import numpy as np
import torch


def make_weights(shapes, seed=0):
    """Deterministic state-dict for a {name: shape} table (checkpoint stand-in; there is no
    network for the real e2e_mask_rcnn_R_50_FPN_1x.pth).  Scales are chosen so that activations
    stay O(1) through the 16 bottlenecks on N(0,50^2) inputs: convs/linears N(0, g*2/fan_in),
    FrozenBN weight U(0.5,1)*k (k=0.4 on the residual-closing bn3), var U(0.8,1.2), mean/bias
    N(0,0.1); predictor layers shrunk so logits are O(1) and box deltas O(0.1)."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        if "cell_anchors" in name:  # derived constants (rpn/anchor_generator.py:47-60), not weights
            continue
        g = torch.Generator().manual_seed(seed * 100003 + i)
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        is_bn = (".bn" in name) or ("downsample.1." in name)
        if is_bn and leaf == "weight":
            k = 0.4 if ".bn3." in name else (0.7 if "downsample.1." in name else 1.0)
            t = (torch.rand(shape, generator=g) * 0.5 + 0.5) * k
        elif is_bn and leaf == "running_var":
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif is_bn:
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.01
            if "mask_fcn_logits" in name:
                t = t + 1.0  # pseudo-masks mostly foreground inside their boxes (non-degenerate MGD)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            if "stem.conv1" in name:
                t = t / 50.0
            if "fpn_inner" in name:
                t = t * 0.5
            if "bbox_pred" in name:
                t = t * 0.1
            if "cls_score" in name or "cls_logits" in name or "mask_fcn_logits" in name:
                t = t * 0.7
        else:
            t = torch.randn(shape, generator=g) * 0.1
        out[name] = t.float()
    return out


def make_labeled(n_img, size, n_inst, seed, device="cpu"):
    """-> images (n,3,size,size) fp32 ~ N(0,50^2); targets: list of dict(boxes (G,4) xyxy fp32,
    labels (G,) int64 alternating 1 (cyto, 16-gon r in U(20,80)*s) / 2 (nuclei, r in U(8,20)*s),
    polys: list of [1-D fp32 tensor]); s = size/1000."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(n_img, 3, size, size, generator=g) * 50.0
    s = size / 1000.0
    targets = []
    for _ in range(n_img):
        boxes, labels, polys = [], [], []
        for k in range(n_inst):
            cls = 1 + (k % 2)
            lo, hi = ((20, 80) if cls == 1 else (8, 20))
            r = (lo + (hi - lo) * torch.rand(1, generator=g).item()) * s
            cx = (100 + 800 * torch.rand(1, generator=g).item()) * s
            cy = (100 + 800 * torch.rand(1, generator=g).item()) * s
            ang = torch.arange(16, dtype=torch.float64) * (2 * math.pi / 16)
            rr = r * (0.85 + 0.3 * torch.rand(16, generator=g).double())
            px = (cx + rr * torch.cos(ang)).clamp(0, size - 1)
            py = (cy + rr * torch.sin(ang)).clamp(0, size - 1)
            poly = torch.stack([px, py], 1).reshape(-1).float()
            polys.append([poly])
            boxes.append([px.min().item(), py.min().item(), px.max().item(), py.max().item()])
            labels.append(cls)
        targets.append(dict(boxes=torch.tensor(boxes, dtype=torch.float32),
                            labels=torch.tensor(labels, dtype=torch.int64), polys=polys,
                            size=(size, size)))
    return imgs.to(device), targets


def make_unlabeled(n_img, size, n_aug, seed, device="cpu"):
    """-> list of n_aug tensors (n,3,size,size): base + N(0,5^2) each (colour-jitter stand-in)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n_img, 3, size, size, generator=g) * 50.0
    return [(base + torch.randn(n_img, 3, size, size, generator=g) * 5.0).to(device) for _ in range(n_aug)]
