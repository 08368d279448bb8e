"""Synthetic head outputs of the proposal / detection stages at sizes where every cap binds (2 images padded to 1024 x 1024:
5 levels of (256, 128, 64, 32, 16)^2 x 3 anchors; 1000 proposals per image through the detection post-processor).  Plain
torch on the CPU, shared by tests/test_proposals_gpu.py (product vs oracle), tests/test_oracle_golden.py (oracle vs fixture)
and tests/golden/gen_golden.py (the REFERENCE's own RPNPostProcessor / PostProcessor on the same inputs -> the fixture).

Scores are drawn WITHOUT ties: objectness logits are logit(p) for distinct p on a 2^-18 grid, so that a last-ulp difference
between two sigmoid implementations cannot reorder two candidates -- the reference leaves the order of equal scores
unspecified."""
import torch

N_IMG, SIZE, PAD = 2, 1000, 1024
GRIDS = [PAD // s for s in (4, 8, 16, 32, 64)]
A = 3


def head_outputs(seed):
    """distinct objectness probabilities per image over all levels; box deltas N(0, 0.5^2) (clip at log(1000/16) active
    for some), NCHW on the host"""
    g = torch.Generator().manual_seed(seed)
    per_img = sum(A * s * s for s in GRIDS)
    grid = 1 << 18
    assert per_img < grid
    obj = [[] for _ in GRIDS]
    for _ in range(N_IMG):
        p = (torch.randperm(grid - 1, generator=g)[:per_img].double() + 1) / grid
        lg = torch.log(p / (1 - p)).float()
        o = 0
        for l, s in enumerate(GRIDS):
            n = A * s * s
            obj[l].append(lg[o:o + n].view(A, s, s))
            o += n
    objectness = [torch.stack(o) for o in obj]
    regression = [torch.randn(N_IMG, 4 * A, s, s, generator=g) * 0.5 for s in GRIDS]
    regression[0][:, 2::4] += 3.0  # widths beyond the clip on level 0
    return objectness, regression


def gt_boxes(seed):
    """12 ground-truth boxes per image (xyxy)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(N_IMG):
        xy = torch.rand(12, 2, generator=g) * 800 + 50
        wh = torch.rand(12, 2, generator=g) * 100 + 10
        out.append(torch.cat([xy, xy + wh], 1))
    return out


def box_head_inputs(seed=8, R=1000):
    """-> per-image proposal boxes (R, 4), their objectness (R,), class logits (N_IMG * R, 3), box deltas (N_IMG * R, 12)"""
    g = torch.Generator().manual_seed(seed)
    boxes, objs = [], []
    for _ in range(N_IMG):
        xy = torch.rand(R, 2, generator=g) * 900
        wh = torch.rand(R, 2, generator=g) * 60 + 6
        boxes.append(torch.cat([xy, (xy + wh).clamp(max=SIZE - 1)], 1))
        objs.append(torch.rand(R, generator=g))
    logits = torch.randn(N_IMG * R, 3, generator=g) * 1.5
    deltas = torch.randn(N_IMG * R, 12, generator=g) * 0.5
    return boxes, objs, logits, deltas
