"""CPU tests (-m "not gpu"): host logic of the product package against the golden vectors, the C-ABI library
(loads, exports every symbol include/mmtpsm.h declares; no compute call without a GPU), fail-loud behaviour, and
the N>1 data-parallel exchange on gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import gold, T, ROOT, PKG


def test_library_exports_every_declared_symbol():
    from maskrcnn_benchmark import _hip
    hdr = open(os.path.join(ROOT, "include", "mmtpsm.h")).read()
    declared = sorted(set(re.findall(r"^int (mmt_[a-z0-9_]+)\(", hdr, flags=re.M)))
    assert declared, "no declarations parsed"
    assert os.path.exists(_hip.LIB_PATH), "libmmtpsm.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_hip.exported_symbols()) == declared  # the Python binding covers the whole ABI
    L.mmt_version.restype = ctypes.c_int
    assert L.mmt_version() == 1


def test_no_cpu_fallback():
    """the product must refuse CPU tensors instead of silently computing something else"""
    from maskrcnn_benchmark import _C, _hip
    with pytest.raises(RuntimeError):
        _hip.conv_forward(torch.zeros(1, 4, 8, 8), torch.zeros(8, 4, 1, 1))
    with pytest.raises(RuntimeError):
        _C.nms(torch.tensor([[0., 0., 1., 1.]]), torch.tensor([1.0]), 0.5)
    with pytest.raises(RuntimeError):
        _C.roi_pool_forward()
    assert _C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0  # reference: empty in -> empty out
    src = ""
    for dp, _, fs in os.walk(os.path.join(PKG, "maskrcnn_benchmark")):
        for f in fs:
            if f.endswith(".py"):
                src += open(os.path.join(dp, f)).read()
    assert "import oracle" not in src and "from oracle" not in src  # the oracle is never on the product path


def test_box_coder_matcher_levelmapper_anchors_golden():
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.matcher import Matcher
    from maskrcnn_benchmark.modeling.poolers import LevelMapper
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
    from maskrcnn_benchmark.structures.image_list import ImageList
    from maskrcnn_benchmark.layers import smooth_l1_loss
    from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import sharpen
    g = gold("small_ops")
    props, refs = T(g["props"]), T(g["refs"])
    for nm, w in (("10", (10., 10., 5., 5.)), ("1", (1., 1., 1., 1.))):
        bc = BoxCoder(w)
        np.testing.assert_array_equal(bc.encode(refs, props).numpy(), g["enc" + nm])
        np.testing.assert_allclose(bc.decode(T(g["codes" + nm]), props).numpy(), g["dec" + nm], rtol=1e-6, atol=1e-4)
    for st, sz in zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512)):
        np.testing.assert_array_equal(generate_anchors(st, (sz,), (0.5, 1.0, 2.0)).float().numpy(), g["cell%d" % st])
    ag = AnchorGenerator((32, 64, 128, 256, 512), (0.5, 1.0, 2.0), (4, 8, 16, 32, 64), 0)
    il = ImageList(torch.zeros(2, 3, 96, 128), [(90, 120), (96, 128)])
    feats = [torch.zeros(2, 1, 24, 32), torch.zeros(2, 1, 12, 16), torch.zeros(2, 1, 6, 8), torch.zeros(2, 1, 3, 4),
             torch.zeros(2, 1, 2, 2)]
    anc = ag(il, feats)
    for i in range(2):
        np.testing.assert_array_equal(torch.cat([a.bbox for a in anc[i]]).numpy(), g["anc_img%d" % i])
        np.testing.assert_array_equal(torch.cat([a.get_field("visibility") for a in anc[i]]).numpy(),
                                      g["vis_img%d" % i].astype(bool))
    iou = boxlist_iou(BoxList(refs[:7], (500, 500)), BoxList(props, (500, 500)))
    np.testing.assert_array_equal(iou.numpy(), g["iou"])
    np.testing.assert_array_equal(Matcher(0.7, 0.3, True)(iou.clone()).numpy(), g["match_rpn"])
    np.testing.assert_array_equal(Matcher(0.5, 0.5, False)(iou.clone()).numpy(), g["match_roi"])
    with pytest.raises(ValueError):
        Matcher(0.5, 0.5)(torch.zeros(0, 5))
    np.testing.assert_array_equal(LevelMapper(2, 5)([BoxList(T(g["lvl_boxes"]), (2000, 2000))]).numpy(), g["lvl"])
    x, y = T(g["sl1_x"]), T(g["sl1_y"])
    assert smooth_l1_loss(x, y, beta=1. / 9, size_average=False).item() == pytest.approx(float(g["sl1_b9"]), rel=1e-6)
    np.testing.assert_allclose(sharpen(T(g["sharp_p"]), 0.5).numpy(), g["sharp"], rtol=1e-6)


def test_boxlist_and_imagelist_api():
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.boxlist_ops import cat_boxlist, remove_small_boxes
    from maskrcnn_benchmark.structures.image_list import to_image_list
    b = BoxList(torch.tensor([[0., 0., 9., 9.], [2., 3., 4., 4.], [-5., 2., 300., 50.]]), (100, 60))
    b.add_field("labels", torch.tensor([1, 2, 1]))
    assert b.area().tolist() == [100., 6., 306. * 49.]
    f = b.transpose(0)
    assert f.bbox[0].tolist() == [90., 0., 99., 9.] and f.get_field("labels").tolist() == [1, 2, 1]
    assert b.convert("xywh").convert("xyxy").bbox.tolist() == b.bbox.tolist()
    c = BoxList(b.bbox.clone(), b.size).clip_to_image(remove_empty=False)
    assert c.bbox[2].tolist() == [0., 2., 99., 50.]
    assert len(remove_small_boxes(b, 3)) == 2
    assert len(cat_boxlist([b, b])) == 6 and cat_boxlist([b, b]).get_field("labels").tolist() == [1, 2, 1, 1, 2, 1]
    assert len(b[torch.tensor([True, False, True])]) == 2
    with pytest.raises(ValueError):
        BoxList(torch.zeros(3, 5), (1, 1))
    il = to_image_list([torch.ones(3, 50, 70), torch.ones(3, 64, 40)], 32)
    assert tuple(il.tensors.shape) == (2, 3, 64, 96) and il.image_sizes == [(50, 70), (64, 40)]
    assert il.tensors[0, :, 50:, :].abs().sum() == 0 and il.tensors[1, :, :, 40:].abs().sum() == 0
    before = il.tensors.clone()
    il.hflip()
    assert torch.equal(il.tensors, torch.flip(before, (3,)))


def test_loss_weighting_schedule_and_ema_alpha_golden():
    from maskrcnn_benchmark.engine.MTtrainer import weight_sum_losses
    from maskrcnn_benchmark.solver.build import WarmupMultiStepLR
    g = gold("mt_losses")
    bal = {"mt_classifier": 0.2, "nms_loss": 1.0, "mt_fg_loss": 1.0}
    for row in g["wsl"]:
        ld = {"loss_classifier": 1.0, "mt_classifier": 1.0, "mt_fg_loss": 1.0, "nms_loss": 1.0}
        w = weight_sum_losses(ld, int(row[0]), 250, 250, 7000, l=5.0, balanced=bal, start_mt=1000)
        np.testing.assert_allclose([w["loss_classifier"], w["mt_classifier"], w["mt_fg_loss"], w["nms_loss"]], row[1:],
                                   rtol=1e-12)

    class Opt:
        lr_factor = 0.0
    o = Opt()
    s = WarmupMultiStepLR(o, (5000,), 0.1, 1.0 / 3, 500, "linear")
    assert o.lr_factor == pytest.approx(1.0 / 3)
    for _ in range(250):
        s.step()
    assert o.lr_factor == pytest.approx(1.0 / 3 * 0.5 + 0.5)
    for _ in range(5000):
        s.step()
    assert o.lr_factor == pytest.approx(0.1)


def test_config_surface():
    from maskrcnn_benchmark.config import cfg, make_default_cfg
    c = make_default_cfg()
    c.merge_from_list(["MT.LAMBDA", 2.5, "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", "123", "MT.FLIP", "False"])
    assert c.MT.LAMBDA == 2.5 and c.MODEL.RPN.PRE_NMS_TOP_N_TRAIN == 123 and c.MT.FLIP is False
    c2 = c.clone()
    c.freeze()
    with pytest.raises(AttributeError):
        c.MT.LAMBDA = 1
    c2.MT.LAMBDA = 1.0
    assert cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS == (10.0, 10.0, 5.0, 5.0)


@pytest.mark.parametrize("irnet", [False, True])
def test_model_state_dict_keys_match_reference(state_shapes, irnet):
    import json, os
    from conftest import GOLD
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = make_default_cfg()
    if irnet:  # BASELINE config 5: relation NMS + mask relation on
        cfg.merge_from_list(["MODEL.RELATION_NMS.USE_RELATION_NMS", True, "MODEL.RELATION_MASK.USE_RELATION", True])
        state_shapes = json.load(open(os.path.join(GOLD, "state_shapes_irnet.json")))
    m = build_detection_model(cfg)
    sd = m.state_dict()
    ref = state_shapes["shapes"]
    assert all(list(v.shape) == ref[k] for k, v in sd.items())
    assert sorted(ref) == sorted(sd)
    order = [k for k, _ in m.named_parameters()]
    assert order == state_shapes["param_order"]  # the EMA zips teacher and student parameters by order
    frozen = {k for k, p in m.named_parameters() if not p.requires_grad}
    ref_frozen = {k for k in order if k not in state_shapes["trainable"]}
    if not irnet:
        # the reference builds mask_relation_module even when it is off (mask_head.py:49); it never receives a
        # gradient there, so torch SGD skips it -- the flat SGD gets the same effect by freezing it
        ref_frozen |= {k for k in order if "mask_relation_module" in k}
    assert frozen == ref_frozen


_DP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "mmt-psm_amd"))
from maskrcnn_benchmark.engine.flat import FlatParams
from maskrcnn_benchmark.engine.MTtrainer import allreduce_gradients, reduce_loss_dict
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Linear(7, 2))
for p in m[0].parameters():
    pass
m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)
flat = FlatParams(m)
assert flat.n_weights > 0 and flat.n_biases > 0 and flat.total >= sum(p.numel() for p in m.parameters())
# every parameter is a view of the flat buffer, every .grad a view of the flat grad buffer
flat.data.add_(1.0)
assert all(float((p.data - 1.0).abs().max()) < 10 for p in m.parameters())
g = torch.Generator().manual_seed(100 + rank)
for p in m.parameters():
    p.grad.copy_(torch.randn(p.shape, generator=g))
mine = flat.grad.clone()
allreduce_gradients(flat)
gathered = [torch.zeros_like(mine) for _ in range(ws)]
dist.all_gather(gathered, mine)
assert torch.allclose(flat.grad, sum(gathered) / ws, atol=1e-6)   # N-rank result == mean of the per-rank gradients
assert all(torch.equal(p.grad.reshape(-1)[:1], p.grad.reshape(-1)[:1]) for p in m.parameters())
red = reduce_loss_dict({"a": torch.tensor(float(rank + 1)), "b": torch.tensor(2.0)})
if rank == 0:
    assert abs(float(red["a"]) - (sum(range(1, ws + 1)) / ws)) < 1e-6 and abs(float(red["b"]) - 2.0) < 1e-6
dist.destroy_process_group()
sys.stdout.write("rank%dok\n" % rank); sys.stdout.flush()
"""


def test_data_parallel_exchange_gloo_world2(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "rank0ok" in out.stdout and "rank1ok" in out.stdout


_BUCKET_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "mmt-psm_amd"))
from torch import nn
from maskrcnn_benchmark.engine.flat import FlatParams
from maskrcnn_benchmark.engine import MTtrainer as MT
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()


class Body(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer1 = nn.Linear(8, 8)
        self.layer2 = nn.Linear(8, 16)
        self.layer3 = nn.Linear(16, 24)
        self.layer4 = nn.Linear(24, 8)
        self.grad_ready = None


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Module()
        self.backbone.body = Body()
        self.backbone.fpn = nn.Linear(8, 40)
        self.rpn = nn.Linear(40, 3)


torch.manual_seed(0)
net = Net()
for p in net.backbone.body.layer1.parameters():
    p.requires_grad_(False)                      # FREEZE_CONV_BODY_AT = 2: no hook ever fires for layer1's piece
flat = FlatParams(net)
b = MT.BucketedAllReduce(flat, net.backbone.body)
assert set(b.pieces) == {"heads", "layer4", "layer3", "layer2", "layer1"}   # round 5: the heads leave on their own, before the FPN
sent = []
orig = MT.BucketedAllReduce._send
def spy(self, lo, hi):
    sent.append((lo, hi))
    return orig(self, lo, hi)
MT.BucketedAllReduce._send = spy


def run(fires_per_stage):
    # two backbone passes were registered on every rank; `fires_per_stage` of them are back-propagated on this rank
    g = torch.Generator().manual_seed(7 + rank)
    flat.grad.copy_(torch.randn(flat.grad.shape, generator=g))
    mine = flat.grad.clone()
    del sent[:]
    b.install()
    for _ in range(2):
        for st in ("heads", "layer4", "layer3", "layer2"):
            net.backbone.body.grad_ready(st, "registered")
    for _ in range(fires_per_stage):
        for st in ("heads", "layer4", "layer3", "layer2"):        # backward order of one pass
            net.backbone.body.grad_ready(st, "fired")
    b.finish()
    parts = [torch.zeros_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine)
    assert torch.allclose(flat.grad, sum(parts) / ws, atol=1e-6), "exchanged gradient != mean"
    seqs = [None] * ws
    dist.all_gather_object(seqs, list(sent))
    assert all(s == seqs[0] for s in seqs), seqs      # the SAME collective sequence on every rank
    spans, pos = sorted(sent), 0
    for lo, hi in spans:                               # exact cover of the flat gradient, nothing twice
        assert lo == pos, spans
        pos = hi
    assert pos == flat.grad.numel()
    return list(sent)


a = run(2)                         # every rank back-propagates both passes
c = run(2 if rank == 0 else 1)     # rank 1 skipped its consistency branch (teacher found no boxes)
d = run(0 if rank == 0 else 2)     # ... or a rank back-propagated nothing through the backbone at all
assert a == c == d
# the padded loss dict reduces with the same keys everywhere
red = MT.reduce_loss_dict({"loss": torch.tensor(1.0), "mt_fg_loss": torch.tensor(float(rank))})
assert sorted(red) == ["loss", "mt_fg_loss"]
# ADVICE r2: the set of parameters FlatSGD updates is the union over ranks of what each rank's backward touched
flat.touched.clear()
flat.touched.update(["rpn.weight", "backbone.fpn.weight"] if rank == 0 else ["rpn.weight", "backbone.body.layer4.bias"])
MT.sync_touched(flat)
assert flat.touched == {"rpn.weight", "backbone.fpn.weight", "backbone.body.layer4.bias"}, flat.touched
# teacher identity checksum: equal buffers pass, a divergent rank is caught on every rank
t = FlatParams(Net())
t.data.copy_(flat.data)
assert MT.check_teacher_identity(t)
if rank == 1:
    t.data[3] += 1e-4
try:
    MT.check_teacher_identity(t)
    raise SystemExit("divergence not detected")
except RuntimeError:
    pass
dist.destroy_process_group()
sys.stdout.write("rank%dok\n" % rank); sys.stdout.flush()
"""


def test_bucketed_allreduce_rank_invariant_gloo_world2(tmp_path):
    """ranks must issue the same collective sequence even when one of them skipped the consistency
    branch; plus the teacher-identity checksum all-reduce of SURVEY 8(e)"""
    script = tmp_path / "bucket.py"
    script.write_text(_BUCKET_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29534", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "rank0ok" in out.stdout and "rank1ok" in out.stdout


def test_flat_sgd_touches_only_parameters_with_gradients():
    """solver/build.py: torch.optim.SGD skips parameters whose grad is None; the flat optimiser updates only the runs of
    the buffer whose parameters reported a gradient since zero_grad"""
    from torch import nn
    from maskrcnn_benchmark.engine.flat import FlatParams
    torch.manual_seed(1)
    m = nn.Sequential(nn.Linear(4, 6), nn.Linear(6, 10), nn.Linear(10, 3), nn.Linear(3, 2))
    f = FlatParams(m)
    assert f.touched == set()
    x = torch.randn(5, 4)
    m[1](m[0](x)).sum().backward()                   # autograd delivers into the pre-set .grad views: seen by the hook
    assert f.touched == {"0.weight", "0.bias", "1.weight", "1.bias"}
    w = f.active_ranges(f.touched, 0, f.n_weights)
    bz = f.active_ranges(f.touched, f.n_weights, f.n_weights + f.n_biases)
    assert w == [(f.index["0.weight"][0], f.index["2.weight"][0])]          # two neighbours merged into one run
    assert bz == [(f.index["0.bias"][0], f.index["2.bias"][0])]
    f.touched.clear()
    m[3](torch.randn(2, 3)).sum().backward()
    assert f.active_ranges(f.touched, 0, f.n_weights) == [(f.index["3.weight"][0], f.n_weights)]
    f.touched.update({"0.weight", "2.weight"})
    r = f.active_ranges(f.touched, 0, f.n_weights)
    assert r == [(f.index["0.weight"][0], f.index["1.weight"][0]), (f.index["2.weight"][0], f.n_weights)]


def test_predicted_exposed_communication_model():
    """engine/MTtrainer.py::predict_exposed_comm (round 6): the xGMI model of SURVEY section 5 applied to a trace's issue times --
    pieces issued early hide behind the backward pass, a piece issued at its end is exposed by its transfer time"""
    from maskrcnn_benchmark.engine.MTtrainer import predict_exposed_comm
    pieces = [{"mbytes": 40.0, "issued_ms": 5.0}, {"mbytes": 60.0, "issued_ms": 9.0}, {"mbytes": 76.0, "issued_ms": 20.0}]
    p = predict_exposed_comm(pieces, backward_end_ms=20.0, world=8, link_gbs=153.0, latency_ms=0.05)
    ring_last = 2 * 7 / 8 * 76e6 / 153e9 * 1e3 + 0.05
    assert abs(p["ring"]["exposed_comm_ms"] - ring_last) < 1e-3          # the two early pieces were done long before
    assert abs(p["direct"]["exposed_comm_ms"] - (2 / 8 * 76e6 / 153e9 * 1e3 + 0.05)) < 1e-3
    # back-to-back pieces queue behind each other on the communicator
    q = predict_exposed_comm([{"mbytes": 100.0, "issued_ms": 0.0}, {"mbytes": 100.0, "issued_ms": 0.0}], 0.0, world=8)
    assert abs(q["ring"]["last_arrival_ms"] - 2 * (2 * 7 / 8 * 100e6 / 153e9 * 1e3 + 0.05)) < 1e-3
    assert predict_exposed_comm(pieces, backward_end_ms=100.0)["ring"]["exposed_comm_ms"] == 0.0
