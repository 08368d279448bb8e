"""Input augmentation on the MI355X against the golden outputs of the reference's own transform classes
(seed for seed: the mirror draws from `random` / `numpy.random` in the reference's order) and against oracle/transforms.py
on random parameters.  Bit-exact: uint8 pixel arithmetic and the fp32 tensor."""
import copy
import random

import numpy as np
import pytest
import torch

from conftest import gold

pytestmark = pytest.mark.gpu


def _cfg(min_size, max_size):
    from maskrcnn_benchmark.config import make_default_cfg
    cfg = make_default_cfg()
    cfg.merge_from_list(["INPUT.MIN_SIZE_TRAIN", min_size, "INPUT.MAX_SIZE_TRAIN", max_size])
    return cfg


def test_transforms_golden_seed_for_seed():
    from maskrcnn_benchmark.data.transforms import build_transforms, DeviceImage
    from maskrcnn_benchmark.data.transforms.transforms import augment_views
    g = gold("transforms")
    cfg = _cfg(80, 133)
    for case in range(2):
        img = g["img%d" % case]
        # unlabeled sample exactly as data/datasets/Pap.py:818-830 drives the two-part transform
        random.seed(100 + case)
        np.random.seed(200 + case)
        tr = build_transforms(cfg, True, "no_label")
        base, _ = tr[0](DeviceImage(img), None)
        for k in range(3):
            t, _ = tr[1](copy.deepcopy(base), None)
            np.testing.assert_array_equal(t.cpu().numpy(), g["no_label%d_view%d" % (case, k)])
        # the same three views in one launch
        random.seed(100 + case)
        np.random.seed(200 + case)
        tr = build_transforms(cfg, True, "no_label")
        base, _ = tr[0](DeviceImage(img), None)
        views = augment_views(base, tr[1], 3)
        for k in range(3):
            np.testing.assert_array_equal(views[k].cpu().numpy(), g["no_label%d_view%d" % (case, k)])
        # labeled sample
        random.seed(100 + case)
        np.random.seed(200 + case)
        t, _ = build_transforms(cfg, True, "source")(DeviceImage(img), None)
        np.testing.assert_array_equal(t.cpu().numpy(), g["source%d" % case])


def test_transforms_against_oracle_random_parameters():
    from oracle import transforms as OT
    from maskrcnn_benchmark import _hip as H
    from maskrcnn_benchmark.data.transforms.transforms import _resample_tables
    rng = np.random.default_rng(21)
    for trial in range(12):
        h, w = int(rng.integers(20, 200)), int(rng.integers(20, 200))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if trial % 3 == 0:
            img[: h // 2] = img[: h // 2] // 32 * 32  # flat regions: grey pixels, saturation 0, ties in max/min
        oh, ow = int(rng.integers(10, 220)), int(rng.integers(10, 220))
        px = torch.from_numpy(img).cuda()
        r = H.resample_u8(px, ow, True, *_resample_tables(w, ow, px.device))
        r = H.resample_u8(r, oh, False, *_resample_tables(h, oh, px.device))
        ref = OT.np_resize(img, oh, ow)
        np.testing.assert_array_equal(r.cpu().numpy(), ref)
        V = 4
        b = rng.uniform(0.85, 1.15, V).astype(np.float32)
        c = rng.uniform(0.85, 1.15, V).astype(np.float32)
        hu = rng.uniform(-0.05, 0.05, V)
        b[0], c[0], hu[0] = 1.0, 1.0, 0.0
        b[1], c[1] = 0.0, 2.0
        flip = bool(trial % 2)
        out = torch.zeros((V, oh + 3, ow + 5, 3), device="cuda")
        shift = np.array([OT.hue_shift(x) for x in hu], np.int32)
        H.aug_views(r, flip, torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda(), torch.from_numpy(shift).cuda(),
                    OT.PIXEL_MEAN, out)
        for v in range(V):
            _, t = OT.view(ref, flip, float(b[v]), float(c[v]), float(hu[v]), [], [], None, True)
            np.testing.assert_array_equal(out[v, :oh, :ow].permute(2, 0, 1).cpu().numpy(), t)
        assert float(out[:, oh:].abs().sum()) == 0 and float(out[:, :, ow:].abs().sum()) == 0  # padding untouched


def test_transforms_full_size_properties():
    """1000 x 1000 crop -> 800 x 800 (the reference's INPUT sizes), 3 views: flip of flip, erased rectangles are exactly the
    drawn bytes, the untouched rest equals the no-erasing view, plain pipeline == bytes - mean"""
    from maskrcnn_benchmark.data.transforms import build_transforms, DeviceImage
    from maskrcnn_benchmark.data.transforms import transforms as T
    from maskrcnn_benchmark.config import make_default_cfg
    cfg = make_default_cfg()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1000, 1000, 3), dtype=np.uint8)
    random.seed(1)
    np.random.seed(2)
    tr = build_transforms(cfg, True, "no_label")
    base, _ = tr[0](DeviceImage(img), None)
    assert base.size == (800, 800)
    views = T.augment_views(base, tr[1], 3, size_divisible=32)
    assert tuple(views.shape) == (3, 3, 800, 800) and views.is_contiguous(memory_format=torch.channels_last)
    # plain pipeline on the resized pixels: channel c of the tensor is byte (2 - c) minus the mean
    plain = T._materialize([T.ToTensor()(base, None)[0]], cfg.INPUT.PIXEL_MEAN)[0]
    px = base.pixels if not base.flip else torch.flip(base.pixels, (1,))
    expect = px.permute(2, 0, 1)[[2, 1, 0]].float() - torch.tensor(cfg.INPUT.PIXEL_MEAN, device="cuda")[:, None, None]
    assert float((plain - expect).abs().max()) <= 1e-4  # (p/255)*255 is p up to one ulp
    # flipping twice is the identity
    a = base.clone()
    a.flip = not a.flip
    pa = T._materialize([T.ToTensor()(a, None)[0]], cfg.INPUT.PIXEL_MEAN)[0]
    assert torch.equal(torch.flip(pa, (2,)), plain)
