"""Golden-vector generator.  BUILD CONTAINER ONLY (needs /root/reference).

Imports the real reference through oracle/refharness/ref_import.py and freezes its outputs on
seeded inputs into small .npz/.pt fixtures next to this file.  The fixtures are DATA (inputs and
expected outputs); no reference source text is stored.

    python tests/golden/gen_golden.py            # regenerates everything (~2 min)
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from oracle.refharness.ref_import import load_reference  # noqa: E402


def _load_synth():
    spec = importlib.util.spec_from_file_location("synthetic", os.path.join(ROOT, "mmt-psm_amd", "synthetic.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()
mb, make_cfg = load_reference()
C = mb._C
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------- native ops
def gen_nms():
    g = torch.Generator().manual_seed(11)
    d = {}
    # case 0: SURVEY App. C known answer
    d["b0"] = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [0, 0, 10, 10.5]])
    d["s0"] = torch.tensor([.5, .9, .3, .8])
    d["t0"] = 0.5
    # case 1: 500 random overlapping boxes, distinct scores
    xy = torch.rand(500, 2, generator=g) * 200
    wh = torch.rand(500, 2, generator=g) * 80 + 4
    d["b1"] = torch.cat([xy, xy + wh], 1)
    d["s1"] = torch.rand(500, generator=g)
    d["t1"] = 0.7
    # case 2: the >= boundary (D9): IoU exactly 0.5 must be suppressed on the CPU path
    d["b2"] = torch.tensor([[0., 0., 9., 9.], [0., 0., 9., 4.], [0., 5., 9., 9.], [20., 20., 29., 29.]])
    d["s2"] = torch.tensor([0.9, 0.8, 0.7, 0.6])
    d["t2"] = 0.5
    # case 3: 2000 boxes score-sorted like the RPN path, thr 0.7
    xy = torch.rand(2000, 2, generator=g) * 400
    wh = torch.rand(2000, 2, generator=g) * 120 + 2
    d["b3"] = torch.cat([xy, xy + wh], 1)
    d["s3"] = torch.sort(torch.rand(2000, generator=g), descending=True)[0]
    d["t3"] = 0.7
    # case 4: single box; case 5: identical boxes with distinct scores
    d["b4"] = torch.tensor([[3., 4., 10., 12.]])
    d["s4"] = torch.tensor([0.3])
    d["t4"] = 0.5
    d["b5"] = torch.tensor([[1., 1., 5., 5.]] * 6)
    d["s5"] = torch.tensor([.1, .6, .3, .5, .2, .4])
    d["t5"] = 0.5
    out = {}
    for i in range(6):
        out["b%d" % i] = d["b%d" % i].float()
        out["s%d" % i] = d["s%d" % i].float()
        out["t%d" % i] = np.float32(d["t%d" % i])
        out["k%d" % i] = C.nms(d["b%d" % i].float(), d["s%d" % i].float(), float(d["t%d" % i]))
    out["k_empty"] = C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5)
    save("nms", **out)


def gen_roi_align():
    g = torch.Generator().manual_seed(12)
    out = {}
    x0 = torch.arange(2 * 3 * 8 * 8).view(2, 3, 8, 8).float()
    r0 = torch.tensor([[0, 0, 0, 7, 7], [1, 2, 2, 5, 6.5]])
    cases = [(x0, r0, 0.5, 2, 2, 2)]
    x1 = torch.randn(2, 16, 40, 48, generator=g)
    r1 = []
    for i in range(40):
        b = i % 2
        x = torch.rand(1, generator=g).item() * 180 - 20
        y = torch.rand(1, generator=g).item() * 150 - 20
        w = torch.rand(1, generator=g).item() * 120
        h = torch.rand(1, generator=g).item() * 100
        r1.append([b, x, y, x + w, y + h])
    r1 += [[0, -50, -50, -40, -45], [1, 500, 500, 600, 600], [0, 10, 10, 10, 10], [1, 0, 0, 191, 159],
           [0, 30.5, 20.25, 31.0, 20.5]]
    r1 = torch.tensor(r1, dtype=torch.float32)
    cases.append((x1, r1, 0.25, 7, 7, 2))
    cases.append((x1, r1, 0.25, 14, 14, 2))
    cases.append((x1, r1, 0.125, 7, 7, 0))  # adaptive sampling grid
    cases.append((x1, r1[:0], 0.25, 7, 7, 2))  # empty
    x2 = torch.randn(1, 4, 5, 7, generator=g)
    r2 = torch.tensor([[0, 0, 0, 100, 100], [0, 2, 1, 3, 2]], dtype=torch.float32)
    cases.append((x2, r2, 1.0 / 16, 3, 5, 2))  # non-square pooled size
    for i, (x, r, sc, ph, pw, sr) in enumerate(cases):
        out["x%d" % i] = x
        out["r%d" % i] = r
        out["p%d" % i] = np.array([sc, ph, pw, sr], dtype=np.float64)
        out["y%d" % i] = C.roi_align_forward(x, r, sc, ph, pw, sr)
    out["n"] = np.int64(len(cases))
    save("roi_align", **out)


def gen_small_ops():
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import generate_anchors, AnchorGenerator
    from maskrcnn_benchmark.modeling.matcher import Matcher
    from maskrcnn_benchmark.modeling.poolers import LevelMapper
    from maskrcnn_benchmark.layers import smooth_l1_loss
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
    from maskrcnn_benchmark.structures.image_list import ImageList
    from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import sharpen
    g = torch.Generator().manual_seed(13)
    out = {}
    xy = torch.rand(64, 2, generator=g) * 300
    wh = torch.rand(64, 2, generator=g) * 150 + 1
    props = torch.cat([xy, xy + wh], 1)
    xy = torch.rand(64, 2, generator=g) * 300
    wh = torch.rand(64, 2, generator=g) * 150 + 1
    refs = torch.cat([xy, xy + wh], 1)
    for nm, w in (("10", (10., 10., 5., 5.)), ("1", (1., 1., 1., 1.))):
        bc = BoxCoder(w)
        enc = bc.encode(refs, props)
        out["enc" + nm] = enc
        codes = torch.randn(64, 12, generator=g) * 2
        codes[0, 2] = 50.0  # exercises bbox_xform_clip
        out["codes" + nm] = codes
        out["dec" + nm] = bc.decode(codes, props)
    out["props"] = props
    out["refs"] = refs
    for st, sz in zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512)):
        out["cell%d" % st] = generate_anchors(st, (sz,), (0.5, 1.0, 2.0)).float()
    out["cell16_9"] = generate_anchors(16, (128, 256, 512), (0.5, 1, 2))  # the in-file KAT table
    ag = AnchorGenerator((32, 64, 128, 256, 512), (0.5, 1.0, 2.0), (4, 8, 16, 32, 64), 0)
    il = ImageList(torch.zeros(2, 3, 96, 128), [(90, 120), (96, 128)])
    feats = [torch.zeros(2, 1, 24, 32), torch.zeros(2, 1, 12, 16), torch.zeros(2, 1, 6, 8),
             torch.zeros(2, 1, 3, 4), torch.zeros(2, 1, 2, 2)]
    anc = ag(il, feats)
    for i in range(2):
        out["anc_img%d" % i] = torch.cat([a.bbox for a in anc[i]], 0)
        out["vis_img%d" % i] = torch.cat([a.get_field("visibility") for a in anc[i]], 0)
    a = BoxList(refs[:7], (500, 500))
    b = BoxList(props, (500, 500))
    iou = boxlist_iou(a, b)
    out["iou"] = iou
    out["match_rpn"] = Matcher(0.7, 0.3, allow_low_quality_matches=True)(iou.clone())
    out["match_roi"] = Matcher(0.5, 0.5, allow_low_quality_matches=False)(iou.clone())
    big = torch.cat([props, props * 4, props * 0.1], 0)
    out["lvl_boxes"] = big
    out["lvl"] = LevelMapper(2, 5)([BoxList(big, (2000, 2000))])
    xx = torch.randn(100, 4, generator=g)
    yy = torch.randn(100, 4, generator=g)
    out["sl1_x"] = xx
    out["sl1_y"] = yy
    out["sl1_b9"] = smooth_l1_loss(xx, yy, beta=1. / 9, size_average=False)
    out["sl1_b1"] = smooth_l1_loss(xx, yy, beta=1, size_average=False)
    p = torch.softmax(torch.randn(10, 3, generator=g), 1)
    out["sharp_p"] = p
    out["sharp"] = sharpen(p, 0.5)
    save("small_ops", **out)


def gen_mt_losses():
    from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import make_roi_box_loss_evaluator
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import fg_hint_loss
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.engine.MTtrainer import weight_sum_losses
    out = {}
    g = torch.Generator().manual_seed(1234)
    # case 4: RANK_FILTER 0 (the yacs default) with 'bce': cls_loss receives the mean of the per-view softmax probabilities
    for case, (R, typ, rf) in enumerate(((12, "bce", 0.2), (257, "bce", 0.2), (257, "kl", 0.2), (64, "mse", 0.2),
                                          (64, "bce", 0.0))):
        cfg = make_cfg(["MT.CLS_LOSS_TYPE", typ, "MT.RANK_FILTER", rf])
        ev = make_roi_box_loss_evaluator(cfg)
        if R == 12:
            labels = torch.tensor([1, 2, 0, 0, 0, 1, 0, 2, 0, 0, 1, 0])
        else:
            labels = (torch.rand(R, generator=g) * 3).long() * (torch.rand(R, generator=g) > 0.5).long()
        t = [torch.randn(R, 3, generator=g) for _ in range(4)]
        s = torch.randn(R, 3, generator=g)
        bl = BoxList(torch.zeros(R, 4), (10, 10))
        bl.add_field("labels", labels)
        out["psm%d_labels" % case] = labels
        out["psm%d_t" % case] = torch.stack(t)
        out["psm%d_s" % case] = s
        out["psm%d" % case] = ev.evaluatePSM([s], [x.clone() for x in t], [bl])
        if case == 0:
            tp = [[torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 4, 4, generator=g)] for _ in range(4)]
            sp = [[torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 4, 4, generator=g)]]
            mk = [(torch.rand(16, 16, generator=g) > 0.5).long() for _ in range(2)]
            for i in range(4):
                out["mgd0_t%d_0" % i] = tp[i][0]
                out["mgd0_t%d_1" % i] = tp[i][1]
            out["mgd0_s_0"] = sp[0][0]
            out["mgd0_s_1"] = sp[0][1]
            out["mgd0_m"] = torch.stack(mk)
            out["mgd0"] = fg_hint_loss(tp, sp, mk)
    # MGD at 5 levels with a non-divisible mask size (the D11 mis-registration path: 100 -> 128 pad)
    g = torch.Generator().manual_seed(77)
    sh = [(2, 8, 32, 32), (2, 8, 16, 16), (2, 8, 8, 8), (2, 8, 4, 4), (2, 8, 2, 2)]
    tp = [[torch.randn(s, generator=g) for s in sh] for _ in range(4)]
    sp = [[torch.randn(s, generator=g) for s in sh]]
    mk = [(torch.rand(100, 100, generator=g) > 0.6).long() * 2 for _ in range(2)]
    for i in range(4):
        for l in range(5):
            out["mgd1_t%d_%d" % (i, l)] = tp[i][l]
    for l in range(5):
        out["mgd1_s_%d" % l] = sp[0][l]
    out["mgd1_m"] = torch.stack(mk)
    out["mgd1"] = fg_hint_loss(tp, sp, mk)
    # loss weighting table (engine/MTtrainer.py:67-109) and EMA alpha (:277-281)
    rows = []
    bal = {"mt_classifier": 0.2, "nms_loss": 1.0, "mt_fg_loss": 1.0}
    for step in (0, 1, 999, 1000, 1001, 1100, 1249, 1250, 3000, 6750, 6751, 6900, 6999, 7000):
        ld = {"loss_classifier": 1.0, "mt_classifier": 1.0, "mt_fg_loss": 1.0, "nms_loss": 1.0}
        w = weight_sum_losses(ld, step, 250, 250, 7000, l=5.0, balanced=bal, start_mt=1000)
        rows.append([step, w["loss_classifier"], w["mt_classifier"], w["mt_fg_loss"], w["nms_loss"]])
    out["wsl"] = np.array(rows, dtype=np.float64)
    # EMA: 21 steps on a small vector through the reference trainer's OWN update_teacher (engine/MTtrainer.py:277-281),
    # called unbound on a stand-in that has the three attributes the method reads
    from types import SimpleNamespace
    from maskrcnn_benchmark.engine.MTtrainer import MTtrainer as RefTrainer
    tp, sp = torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(5))
    holder = SimpleNamespace(alpha=0.99, teacher=SimpleNamespace(parameters=lambda: [tp]),
                             student=SimpleNamespace(parameters=lambda: [sp]))
    s0 = torch.arange(5).float()
    tr = []
    for it in range(21):
        sp.data.copy_(s0 * (1 + 0.1 * it))
        RefTrainer.update_teacher(holder, it)
        tr.append(tp.data.clone())
    out["ema_trace"] = torch.stack(tr)
    save("mt_losses", **out)


def gen_masks():
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.inference import paste_mask_in_image
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.loss import project_masks_on_boxes
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    out = {}
    g = torch.Generator().manual_seed(21)
    masks = torch.rand(12, 28, 28, generator=g)
    boxes = torch.tensor([[10.3, 12.7, 60.2, 80.9], [-5.0, -3.0, 20.0, 30.0], [100.0, 90.0, 140.5, 127.9],
                          [50.0, 50.0, 50.5, 50.5], [0.0, 0.0, 149.0, 127.0], [70.2, 3.3, 90.9, 33.1],
                          [120.0, 100.0, 170.0, 150.0], [33.3, 44.4, 55.5, 66.6], [5.5, 5.5, 9.5, 9.5],
                          [60.0, 60.0, 61.0, 100.0], [1.0, 100.0, 140.0, 110.0], [140.0, 2.0, 149.0, 20.0]])
    out["paste_masks"] = masks
    out["paste_boxes"] = boxes
    out["paste_out"] = torch.stack([paste_mask_in_image(m, b, 128, 150, 0.5, 1) for m, b in zip(masks, boxes)])
    imgs, tg = synth.make_labeled(1, 200, 8, seed=5)
    t = tg[0]
    polys = [[p.tolist() for p in inst] for inst in t["polys"]]
    # also a two-polygon instance (rleMerge union path)
    polys.append([polys[0][0], polys[1][0]])
    seg = SegmentationMask(polys, (200, 200), mode="poly")
    props = torch.cat([t["boxes"], t["boxes"][:1]], 0).clone()
    props += torch.randn(props.shape, generator=g) * 3
    props = props.clamp(0, 199)
    bl = BoxList(props, (200, 200))
    out["proj_boxes"] = props
    out["proj_npoly"] = np.array([len(p) for p in polys])
    flat = []
    for inst in polys:
        for p in inst:
            flat.append(np.asarray(p, dtype=np.float32))
    out["proj_polylens"] = np.array([len(p) for p in flat])
    out["proj_polys"] = np.concatenate(flat)
    out["proj_out"] = project_masks_on_boxes(seg, bl, 28)
    save("masks", **out)


# ------------------------------------------------------------------------------- model level
def to_ref_targets(tgs):
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    res = []
    for t in tgs:
        bl = BoxList(t["boxes"], t["size"], mode="xyxy")
        bl.add_field("labels", t["labels"])
        bl.add_field("masks", SegmentationMask([[p.tolist() for p in inst] for inst in t["polys"]], t["size"], mode="poly"))
        res.append(bl)
    return res


def gen_model(size=160, n_inst=4, tag="model160", relation=False):
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.image_list import to_image_list
    cfg = make_cfg(relation=relation)
    torch.manual_seed(0)
    student = build_detection_model(cfg, is_student=True)
    teacher = build_detection_model(cfg, is_teacher=True)
    shapes = {k: tuple(v.shape) for k, v in student.state_dict().items()}
    with open(os.path.join(HERE, "state_shapes_irnet.json" if relation else "state_shapes.json"), "w") as f:
        json.dump({"shapes": {k: list(v) for k, v in shapes.items()},
                   "param_order": [k for k, _ in student.named_parameters()],
                   "trainable": [k for k, p in student.named_parameters() if p.requires_grad]}, f)
    sd = synth.make_weights(shapes, seed=0)
    student.load_state_dict(sd, strict=False)
    teacher.load_state_dict(sd, strict=False)
    student.train()
    teacher.eval()
    imgs, tgs = synth.make_labeled(2, size, n_inst, seed=1234)
    unl = synth.make_unlabeled(2, size, 3, seed=4321)
    out = {}
    torch.manual_seed(99)
    il = to_image_list(list(imgs), 32)
    ld = student(il, to_ref_targets(tgs))
    for k, v in ld.items():
        out["sup_" + k] = v.detach()
    torch.manual_seed(100)
    t_list = [to_image_list(list(u), 32) for u in unl[:2]]
    s_list = [to_image_list(list(u), 32) for u in unl[-1:]]
    with torch.no_grad():
        tr = teacher.forward_teacher(t_list)
    for i, r in enumerate(tr["result_t"]):
        out["t_res%d_bbox" % i] = r.bbox
        out["t_res%d_labels" % i] = r.get_field("labels")
        out["t_res%d_objectness" % i] = r.get_field("objectness")
    out["t_logits"] = torch.stack(tr["class_logit_t"])
    for i, e in enumerate(tr["embedding"]):
        for l, m in enumerate(e):
            out["t_emb%d_%d_stats" % (i, l)] = torch.stack([m.mean(), m.abs().mean(), m[0, 0, 0, 0], m[-1, -1, -1, -1]])
    out["t_seg"] = torch.stack(tr["seg_mask"]).to(torch.int16)
    torch.manual_seed(101)
    sl = student.forward_student(s_list, tr)
    for k, v in sl.items():
        out["stu_" + k] = v.detach()
    save(tag, **out)
    print({k: float(v) for k, v in out.items() if v.numel() == 1})


def gen_transforms():
    """input augmentation: the reference's OWN transform classes (data/transforms/{transforms,build}.py),
    loaded from the reference tree, with seeded `random` / `numpy.random`.  torchvision is absent from this image: the five
    PIL code paths of torchvision.transforms.functional the classes call are provided by a stand-in module that forwards to
    Pillow (what torchvision itself does for PIL images)."""
    import copy
    import random
    import types
    from PIL import Image, ImageEnhance

    Fm = types.ModuleType("torchvision.transforms.functional")
    Fm.resize = lambda img, size, interpolation=Image.BILINEAR: img.resize(size[::-1], interpolation)
    Fm.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
    Fm.adjust_brightness = lambda img, f: ImageEnhance.Brightness(img).enhance(f)
    Fm.adjust_contrast = lambda img, f: ImageEnhance.Contrast(img).enhance(f)

    def adjust_hue(img, hue_factor):
        h, s, v = img.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        nh += np.uint8(int(hue_factor * 255) & 255)  # `np_h += np.uint8(hue_factor * 255)` with the C wrap-around cast
        return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")

    Fm.adjust_hue = adjust_hue
    Fm.to_tensor = lambda img: torch.from_numpy(np.array(img)).permute(2, 0, 1).float().div(255)
    Fm.normalize = lambda t, mean, std: (t - torch.tensor(mean, dtype=t.dtype)[:, None, None]) / torch.tensor(std, dtype=t.dtype)[:, None, None]
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tv.transforms, tvt.functional = tvt, Fm
    saved = {k: sys.modules.get(k) for k in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional")}
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": Fm})
    try:
        d = os.path.join(os.path.dirname(mb.__file__), "data", "transforms")
        spec = importlib.util.spec_from_file_location("ref_transforms", os.path.join(d, "__init__.py"),
                                                      submodule_search_locations=[d])
        rt = importlib.util.module_from_spec(spec)
        sys.modules["ref_transforms"] = rt
        spec.loader.exec_module(rt)
        cfg = make_cfg(["INPUT.MIN_SIZE_TRAIN", 80, "INPUT.MAX_SIZE_TRAIN", 133])
        out = {}
        rng = np.random.default_rng(5)
        for case, (h, w) in enumerate([(100, 100), (90, 120)]):
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            img[:10, :10] = 0
            img[10:20, :10] = 255
            img[20:30, :10] = (200, 30, 30)
            out["img%d" % case] = img
            for domain in ("no_label", "source"):
                random.seed(100 + case)
                np.random.seed(200 + case)
                tr = rt.build_transforms(cfg, True, domain)
                pil = Image.fromarray(img, "RGB")
                if domain == "no_label":  # data/datasets/Pap.py:818-830
                    base, _ = tr[0](pil, None)
                    for k in range(3):
                        t, _ = tr[1](copy.deepcopy(base), None)
                        out["%s%d_view%d" % (domain, case, k)] = t.numpy()
                else:
                    t, _ = tr(pil, None)
                    out["%s%d" % (domain, case)] = t.numpy()
        save("transforms", **out)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def gen_checkpoint():
    """the reference's own suffix-matching aligner on three key sets, and a tiny checkpoint FILE written by the
    reference's Checkpointer.save (model + torch SGD + the reference's WarmupMultiStepLR) -> tests/golden/checkpoint_*"""
    import tempfile
    import types
    from collections import OrderedDict
    if not hasattr(torch, "_six"):  # utils/imports.py:4 reads torch._six.PY3
        torch._six = types.SimpleNamespace(PY3=True, PY37=True)
    import importlib; hub = importlib.import_module("torch.hub")
    for missing in ("_download_url_to_file", "urlparse", "HASH_REGEX"):  # utils/model_zoo.py:5-7 (URL cache, unused here)
        if not hasattr(hub, missing):
            setattr(hub, missing, None)
    from maskrcnn_benchmark.utils.model_serialization import align_and_update_state_dicts, strip_prefix_if_present
    from maskrcnn_benchmark.utils.checkpoint import Checkpointer
    from maskrcnn_benchmark.solver.lr_scheduler import WarmupMultiStepLR
    shapes = json.load(open(os.path.join(HERE, "state_shapes.json")))["shapes"]
    model_keys = list(shapes.keys())
    cases = {}
    # (a) DataParallel prefix on every loaded key  (b) ImageNet-style: body keys without 'backbone.body.', plus decoys that
    # are SHORTER suffixes of the same keys  (c) the complex model of the reference's tests/checkpoint.py:20-37
    cases["module_prefix"] = (model_keys, ["module." + k for k in model_keys])
    body = [k[len("backbone.body."):] for k in model_keys if k.startswith("backbone.body.")]
    decoys = sorted({k.split(".", 2)[-1] for k in body if k.count(".") >= 2})
    cases["suffix_longest"] = (model_keys, body + decoys + ["fc1000.weight", "unrelated.bias"])
    cases["complex_model"] = (["block1.layer1.weight", "block1.layer1.bias", "layer2.weight", "layer2.bias",
                               "res.layer2.weight", "res.layer2.bias"],
                              ["layer1.weight", "layer1.bias", "layer2.weight", "layer2.bias", "res.layer2.weight",
                               "res.layer2.bias"])
    out = {}
    for name, (cur, loaded) in cases.items():
        msd = OrderedDict((k, torch.tensor([-1.0])) for k in cur)
        lsd = OrderedDict((k, torch.tensor([float(i)])) for i, k in enumerate(loaded))
        lsd = strip_prefix_if_present(lsd, prefix="module.")
        lk = list(lsd.keys())
        vals = {float(v): k for k, v in lsd.items()}
        align_and_update_state_dicts(msd, lsd)
        out[name] = {"model_keys": cur, "loaded_keys": loaded,
                     "mapping": {k: (vals[float(v)] if float(v) >= 0 else None) for k, v in msd.items()}}
    with open(os.path.join(HERE, "checkpoint_align.json"), "w") as f:
        json.dump(out, f)
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Linear(2, 3), torch.nn.Linear(3, 1))
    opt = torch.optim.SGD([{"params": [p], "lr": 0.01, "weight_decay": 1e-4} for p in m.parameters()], 0.01, momentum=0.9)
    m(torch.ones(1, 2)).sum().backward()
    opt.step()
    sched = WarmupMultiStepLR(opt, (5000,), 0.1, warmup_factor=1.0 / 3, warmup_iters=500, warmup_method="linear")
    with tempfile.TemporaryDirectory() as d:
        Checkpointer(torch.nn.DataParallel(m), opt, sched, d, True).save("model_0000007", iteration=7)
        tag = open(os.path.join(d, "last_checkpoint")).read()
        assert tag == os.path.join(d, "model_0000007.pth")
        data = open(os.path.join(d, "model_0000007.pth"), "rb").read()
    with open(os.path.join(HERE, "checkpoint_ref_tiny.pth"), "wb") as f:
        f.write(data)
    print("wrote checkpoint_align.json, checkpoint_ref_tiny.pth (%d bytes)" % len(data),
          {k: sum(v is not None for v in c["mapping"].values()) for k, c in out.items()})


def gen_proposals():
    """VERDICT r2 (next 6a): the REFERENCE's RPNPostProcessor (training selector with GT boxes, TEST-config selector,
    TRAIN-config selector of a teacher in eval mode) and PostProcessor on the head outputs of tests/proposal_inputs.py at
    2 x 1024^2 -- pre-NMS top-2000 on three levels, batch-wide top-2000, per-image top-1000 / 2000, > 400 detections per image
    before the kthvalue cut to 200: the sizes where every cap binds, which the 160^2 model fixtures never reach."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import proposal_inputs as pi
    from maskrcnn_benchmark.modeling.rpn.rpn import RPNModule
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import make_roi_box_post_processor
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import ImageList
    cfg = make_cfg()
    out = {}
    il = ImageList(torch.zeros(pi.N_IMG, 3, pi.PAD, pi.PAD), [(pi.SIZE, pi.SIZE)] * pi.N_IMG)
    feats = [torch.zeros(pi.N_IMG, 1, s, s) for s in pi.GRIDS]

    def dump(tag, lists, fields, int_fields=()):
        out[tag + "_count"] = torch.tensor([len(b) for b in lists])
        for i, b in enumerate(lists):
            out["%s_%d_bbox" % (tag, i)] = b.bbox
            for f in fields:
                out["%s_%d_%s" % (tag, i, f)] = b.get_field(f).float()
            for f in int_fields:
                out["%s_%d_%s" % (tag, i, f)] = b.get_field(f).to(torch.int32)

    with torch.no_grad():
        # (1) training selector of a student RPN: pre-NMS 2000 / level, NMS 0.7, batch-wide top-2000, + GT boxes
        m = RPNModule(cfg, is_teacher=False)
        anchors = m.anchor_generator(il, feats)
        obj, reg = pi.head_outputs(3)
        tg = [BoxList(b, (pi.SIZE, pi.SIZE), mode="xyxy") for b in pi.gt_boxes(5)]
        for t in tg:
            t.add_field("labels", torch.ones(12, dtype=torch.int64))
        m.train()
        dump("train", m.box_selector_train(anchors, obj, reg, tg), ("objectness",))
        # (2) the two selectors a teacher runs on pyramid 0 (generalized_rcnn.py:126,146), eval mode
        t = RPNModule(cfg, is_teacher=True)
        t.eval()
        obj, reg = pi.head_outputs(4)
        dump("test", t.box_selector_test(anchors, obj, reg), ("objectness",))
        dump("teach", t.box_selector_train(anchors, obj, reg, None), ("objectness", "box_reg"), ("rpn_topk", "rpn_ancher_level"))
        # (3) detection post-processor: 1000 proposals per image, kthvalue cut to 200 -- and uncut
        boxes, objs, logits, deltas = pi.box_head_inputs(8, 1000)
        for tag, dets in (("det", 200), ("det_uncut", 10 ** 6)):
            c2 = make_cfg()
            c2.MODEL.ROI_HEADS.DETECTIONS_PER_IMG = dets
            pp = make_roi_box_post_processor(c2)
            props = []
            for b, o in zip(boxes, objs):
                p = BoxList(b.clone(), (pi.SIZE, pi.SIZE), mode="xyxy")
                p.add_field("objectness", o)
                props.append(p)
            dump(tag, pp((logits, deltas), props), ("scores",), ("labels",))
    for k in list(out):
        if k.endswith("_bbox") or k.endswith("_box_reg"):
            out[k] = out[k].float()
    save("proposals1024", **out)
    print({k: out[k].tolist() for k in out if k.endswith("_count")})


def gen_pap_eval():
    """SURVEY 8f-4 / VERDICT r2 (next 9): the REFERENCE's PAP evaluator (data/datasets/evaluation/pap/pap_eval.py: Papeval
    evaluate / accumulate / summarize) with the reference's own pycocotools (pycoco/, built by oracle/refharness) on the
    synthetic windows of tests/pap_inputs.py: RLE strings, the iouIntUni triples of one window and the final statistics."""
    import re
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pap_inputs
    from pycocotools import mask as mu
    # the evaluator module is loaded by path (the package's __init__ pulls in the data loaders: torch._six); its visualisation
    # import is a private module of the authors; numpy 2 refuses the float `num` of two np.linspace calls
    v = types.ModuleType("maskrcnn_benchmark.utils.visual")
    v.display_instance = None
    sys.modules["maskrcnn_benchmark.utils.visual"] = v
    path = os.path.join(os.path.dirname(mb.__file__), "data", "datasets", "evaluation", "pap", "pap_eval.py")
    src = open(path).read()
    src = re.sub(r"np\.round\(\(0\.95 - \.5\) / \.05\) \+ 1", "int(np.round((0.95 - .5) / .05)) + 1", src)
    src = re.sub(r"np\.round\(\(1\.00 - \.0\) / \.01\) \+ 1", "int(np.round((1.00 - .0) / .01)) + 1", src)
    pe = types.ModuleType("ref_pap_eval")
    exec(compile(src, path, "exec"), pe.__dict__)
    gts, dts = pap_inputs.make(7)
    rles = []
    for lst in (gts, dts):
        for x in lst:
            r = mu.encode(np.asfortranarray(x["mask"]))
            x["segmentation"] = {"size": r["size"], "counts": r["counts"].decode("ascii")}
            rles.append(x["segmentation"]["counts"])
    ev = pe.Papeval([{k: v_ for k, v_ in g.items() if k != "mask"} for g in gts],
                    [{k: v_ for k, v_ in d.items() if k != "mask"} for d in dts], "segm")
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    out = {"stats": {m: {str(k): (float(np.asarray(v_).reshape(-1)[0]) if not isinstance(v_, (int, float)) else float(v_))
                         for k, v_ in d.items()} for m, d in ev.stats.items()},
           "rle_counts": rles, "areas": [int(mu.area(x["segmentation"])) for x in gts + dts]}
    # one window's raw iouIntUni triple (detections in score order x ground truths), cells the C code writes
    key = sorted(k for k in ev.ious if len(ev.ious[k]) == 5 and len(ev.ious[k][0]))[0]
    g_ = ev._gts[key]
    d_ = sorted(ev._dts[key], key=lambda q: -q["score"])
    iou, inter, uni = mu.iouIntUni([q["segmentation"] for q in d_], [q["segmentation"] for q in g_], [0] * len(g_))
    out["window"] = {"key": [key[0], key[1]], "iou": iou.tolist(), "inter": np.where(iou > 0, inter, 0).tolist(),
                     "union": np.where(iou > 0, uni, 0).tolist()}
    out["per_window"] = [None if e is None else {"image_id": e["image_id"], "category_id": e["category_id"], "AJI": float(e["AJI"][0, 0]),
                                                  "F1": float(e["F1"]), "FNRo": float(e["FNRo"]), "FDR": float(e["FDR"]),
                                                  "DSC": [float(x) for x in e["DSC"]], "TPRp": [float(x) for x in e["TPRp"]]}
                         for e in ev.evalImgs]
    out["precision_shape"] = list(ev.eval["precision"].shape)
    out["precision_sum"] = float(ev.eval["precision"].sum())
    out["recall"] = ev.eval["recall"].tolist()
    with open(os.path.join(HERE, "pap_eval.json"), "w") as f:
        json.dump(out, f, default=lambda o: o.item() if hasattr(o, "item") else str(o))
    print("wrote pap_eval.json", {m: d for m, d in out["stats"].items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["nms", "roi", "small", "mt", "masks", "model"]
    if "nms" in which:
        gen_nms()
    if "roi" in which:
        gen_roi_align()
    if "small" in which:
        gen_small_ops()
    if "mt" in which:
        gen_mt_losses()
    if "masks" in which:
        gen_masks()
    if "model" in which:
        gen_model()
    if "irnet" in which:
        gen_model(tag="model160_irnet", relation=True)
    if "checkpoint" in which:
        gen_checkpoint()
    if "transforms" in which:
        gen_transforms()
    if "proposals" in which:
        gen_proposals()
    if "pap" in which:
        if len(which) > 1:
            # D14: the reference's rleIouInterUnion leaves the intersection / union cells of pairs whose boxes do
            # not overlap UNWRITTEN (pycoco/maskApi.c:238-259) and caclulateMetrics divides the whole arrays (pap_eval.py:425-477), so
            # per-window DSC / TPRp / FNRo / FDR depend on what the allocator hands out.  In a fresh process that memory is zero pages
            # -- the realisation the fixture (and the product's evaluator, by contract) pins; after other generators it is not.
            import subprocess
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "pap"])
        else:
            gen_pap_eval()
