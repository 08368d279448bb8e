"""checkpoint I/O (reference utils/checkpoint.py:13-205, utils/model_serialization.py:10-80).

The suffix-matching aligner is pinned to the REFERENCE's own output on three key sets (tests/golden/checkpoint_align.json,
written by gen_golden.py from the imported reference), a checkpoint FILE written by the reference's Checkpointer.save
(tests/golden/checkpoint_ref_tiny.pth) is read back, and the reference's own unit tests (tests/checkpoint.py:39-114) are
mirrored against this build's Checkpointer.  CPU only: no kernels involved."""
import json
import os
from collections import OrderedDict

import pytest
import torch
from torch import nn

from conftest import GOLD


def _ckpt():
    from maskrcnn_benchmark.utils import checkpoint, model_serialization
    return checkpoint, model_serialization


def test_aligner_matches_reference_mapping():
    _, ms = _ckpt()
    cases = json.load(open(os.path.join(GOLD, "checkpoint_align.json")))
    assert set(cases) == {"module_prefix", "suffix_longest", "complex_model"}
    for name, c in cases.items():
        msd = OrderedDict((k, torch.tensor([-1.0])) for k in c["model_keys"])
        lsd = OrderedDict((k, torch.tensor([float(i)])) for i, k in enumerate(c["loaded_keys"]))
        lsd = ms.strip_prefix_if_present(lsd, prefix="module.")
        back = {float(v): k for k, v in lsd.items()}
        ms.align_and_update_state_dicts(msd, lsd)
        got = {k: (back[float(v)] if float(v) >= 0 else None) for k, v in msd.items()}
        assert got == c["mapping"], name
        assert sum(v is not None for v in got.values()) > 0


def _model():
    return nn.Sequential(nn.Linear(2, 3), nn.Linear(3, 1))


@pytest.mark.parametrize("wrap_trained,wrap_fresh", [(False, False), (True, False), (False, True), (True, True)])
def test_round_trip_like_reference_tests(tmp_path, wrap_trained, wrap_fresh):
    """tests/checkpoint.py:39-101 of the reference: same folder (last_checkpoint tag) and named file, with and without the
    DataParallel 'module.' prefix on either side"""
    ck, _ = _ckpt()
    trained = nn.DataParallel(_model()) if wrap_trained else _model()
    f = str(tmp_path / "a")
    os.makedirs(f)
    ck.Checkpointer(trained, save_dir=f, save_to_disk=True).save("checkpoint_file")
    fresh = nn.DataParallel(_model()) if wrap_fresh else _model()
    c = ck.Checkpointer(fresh, save_dir=f)
    assert c.has_checkpoint() and c.get_checkpoint_file() == os.path.join(f, "checkpoint_file.pth")
    c.load()
    for a, b in zip(trained.parameters(), fresh.parameters()):
        assert a is not b and a.equal(b)
    g = str(tmp_path / "b")
    os.makedirs(g)
    fresh2 = nn.DataParallel(_model()) if wrap_fresh else _model()
    c2 = ck.Checkpointer(fresh2, save_dir=g)
    assert not c2.has_checkpoint() and c2.get_checkpoint_file() == ""
    assert c2.load() == {}                                  # nothing to load: the model keeps its initialisation
    c2.load(os.path.join(f, "checkpoint_file.pth"))
    for a, b in zip(trained.parameters(), fresh2.parameters()):
        assert a.equal(b)
    ck.Checkpointer(trained, save_dir="", save_to_disk=True).save("nowhere")   # no directory: silently nothing


def test_complex_model_loaded_by_suffix():
    """tests/checkpoint.py:103-114"""
    _, ms = _ckpt()
    for dp in (False, True):
        m = nn.Module()
        m.block1 = nn.Module()
        m.block1.layer1 = nn.Linear(2, 3)
        m.layer2 = nn.Linear(3, 2)
        m.res = nn.Module()
        m.res.layer2 = nn.Linear(3, 2)
        sd = OrderedDict((k, torch.rand(s)) for k, s in (("layer1.weight", (3, 2)), ("layer1.bias", (3,)),
                                                          ("layer2.weight", (2, 3)), ("layer2.bias", (2,)),
                                                          ("res.layer2.weight", (2, 3)), ("res.layer2.bias", (2,))))
        model = nn.DataParallel(m) if dp else m
        ms.load_state_dict(model, sd)
        for loaded, stored in zip(model.state_dict().values(), sd.values()):
            assert loaded is not stored and loaded.equal(stored)


def test_reads_a_checkpoint_written_by_the_reference(tmp_path):
    ck, _ = _ckpt()
    path = os.path.join(GOLD, "checkpoint_ref_tiny.pth")
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"model", "optimizer", "scheduler", "iteration"} and all(k.startswith("module.") for k in raw["model"])
    m = _model()
    extra = ck.Checkpointer(m).load(path, test=True)        # train_mean_teacher.py:42-43
    assert set(extra) == {"optimizer", "scheduler", "iteration"}
    for k, v in m.state_dict().items():
        assert v.equal(raw["model"]["module." + k])
    m2 = _model()
    extra = ck.Checkpointer(m2, save_dir=str(tmp_path)).load(path)   # the fork drops optimizer and scheduler (:90)
    assert extra == {"iteration": 7}
    assert all(a.equal(b) for a, b in zip(m.parameters(), m2.parameters()))


def test_transfer_learning_keeps_fresh_predictors(tmp_path):
    """utils/checkpoint.py:75-78,148-160: a file called e2e_mask_rcnn_R_50_FPN_1x.pth is a pre-trained model of another
    class count -- cls_score / bbox_pred / mask_fcn_logits are not taken from it, iteration = -1"""
    ck, _ = _ckpt()

    def net():
        m = nn.Module()
        m.body = nn.Linear(4, 4)
        m.cls_score = nn.Linear(4, 3)
        m.bbox_pred = nn.Linear(4, 12)
        return m

    torch.manual_seed(1)
    src, dst = net(), net()
    path = str(tmp_path / "e2e_mask_rcnn_R_50_FPN_1x.pth")
    torch.save({"model": src.state_dict(), "optimizer": {}, "scheduler": {}}, path)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    extra = ck.Checkpointer(dst, save_dir=str(tmp_path)).load(path)
    assert extra == {"iteration": -1}
    sd = dst.state_dict()
    assert sd["body.weight"].equal(src.state_dict()["body.weight"])
    for k in ("cls_score.weight", "bbox_pred.bias"):
        assert sd[k].equal(before[k]) and not sd[k].equal(src.state_dict()[k])


def test_detector_checkpoint_round_trip_and_optimizer_state(tmp_path, weights):
    """the whole R50-FPN detector through Checkpointer.save / load with the reference's key names (fc6 in the reference's
    column order on disk), plus the flat optimiser's and the schedule's state dicts"""
    ck, _ = _ckpt()
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.solver import make_optimizer, make_lr_scheduler
    cfg = make_default_cfg()
    a = build_detection_model(cfg, is_student=True)
    a.load_state_dict(weights, strict=False)
    opt = make_optimizer(cfg, a)
    sched = make_lr_scheduler(cfg, opt)
    opt.flat.momentum.copy_(torch.randn(opt.flat.momentum.shape))
    for _ in range(37):
        sched.step()
    c = ck.DetectronCheckpointer(cfg, a, opt, sched, str(tmp_path), True)
    c.save("model_0000050", iteration=50)
    on_disk = torch.load(os.path.join(str(tmp_path), "model_0000050.pth"), map_location="cpu")
    for k in ("box_heads.box.feature_extractor.fc6.weight", "backbone.body.layer3.2.conv2.weight", "rpn.head.conv.bias"):
        assert on_disk["model"][k].equal(weights[k]), k    # reference layout and names on disk
    b = build_detection_model(cfg, is_student=True)
    opt_b = make_optimizer(cfg, b)
    sched_b = make_lr_scheduler(cfg, opt_b)
    extra = ck.DetectronCheckpointer(cfg, b, opt_b, sched_b, str(tmp_path), False).load()   # picks up last_checkpoint
    assert extra == {"iteration": 50}
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and all(sa[k].equal(sb[k]) for k in sa)
    # the fork never restores these two on load(); their state dicts still round-trip for callers that do it themselves
    opt_b.load_state_dict(on_disk["optimizer"])
    sched_b.load_state_dict(on_disk["scheduler"])
    for n, (o, k) in opt.flat.index.items():              # every parameter's slot (the padding between slots is not state)
        if o < opt.flat.n_trainable:
            assert opt_b.flat.momentum[o:o + k].equal(opt.flat.momentum[o:o + k]), n
    assert sched_b.last_epoch == sched.last_epoch
    assert opt_b.lr_factor == pytest.approx(sched.factor())
    with pytest.raises(NotImplementedError):
        c._load_file("catalog://ImageNetPretrained/MSRA/R-50")
