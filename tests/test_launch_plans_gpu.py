"""Launch plans (maskrcnn_benchmark/_hip.py: planned): a no-grad backbone pass is recorded once per shape -- every C-ABI call with
its arguments, its tensors in a private memory pool, its statistics slots in a block of its own -- and replayed afterwards with
only the input's address patched.  The replay must be the pass: bit-identical pyramids for inputs it has never seen, valid
statistics, one plan per shape, nothing recorded when gradients are on."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
sys.path.insert(0, ROOT)


@pytest.fixture()
def det():
    import synthetic
    from maskrcnn_benchmark import _hip as H
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.engine.flat import flatten_model
    H.lib()
    prev = H.get_conv_precision()
    H.set_conv_precision(3)
    H.set_f16x2(True)
    m = build_detection_model(make_default_cfg(), is_teacher=True)
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_shapes.json")))["shapes"]
    m.load_state_dict(synthetic.make_weights(shapes, seed=0), strict=False)
    m.cuda().eval()
    flatten_model(m).refresh_planes()
    H._LAUNCH_PLANS.clear()
    yield H, m
    H.LAUNCH_PLANS = True
    H._LAUNCH_PLANS.clear()
    H.set_f16x2(None)
    H.set_conv_precision(prev)


@pytest.mark.parametrize("c_replay", [True, False])
def test_replayed_backbone_pass_is_the_pass(det, c_replay, monkeypatch):
    """c_replay: the recorded calls go out through ONE library call per run (include/mmtpsm.h: mmt_replay, the interpreter lock
    released) or one interpreter call each"""
    H, m = det
    monkeypatch.setattr(H, "C_REPLAY", c_replay)
    g = torch.Generator().manual_seed(1)
    xs = [(torch.randn(4, 3, 256, 320, generator=g) * 50.0).cuda() for _ in range(5)]
    with torch.no_grad():
        H.LAUNCH_PLANS = False
        want = [tuple(t.clone() for t in m.run_backbone(x)) for x in xs]
        H.LAUNCH_PLANS = True
        calls = []
        for i, x in enumerate(xs):
            c0 = H.C_CALLS[0]
            got = m.run_backbone(x)
            calls.append(H.C_CALLS[0] - c0)
            assert len(got) == len(want[i]) == 5
            for a, b in zip(got, want[i]):
                assert a.shape == b.shape and torch.equal(a, b), (i, (a - b).abs().max().item())
            torch.cuda.synchronize()
            for a in got[:4]:                                  # the statistics a consumer's scale comes from
                slot = a._mmt_amax[0]
                assert a._mmt_amax[1] == a._version
                assert float(torch.as_tensor(0.0)) == 0.0 and abs(float(a.abs().max())) >= 0
                buf = slot.pool.slot_buf if hasattr(slot.pool, "slot_buf") else slot.pool.dev
                assert float(buf[slot.idx, 0]) == float(a.abs().max())
    plans = [p for p in H._LAUNCH_PLANS.values()]
    assert len(plans) == 1 and plans[0].seen == 5 and len(plans[0].calls) > 50
    if c_replay:   # nearly everything went through mmt_replay: a handful of runs, the ATen closures / by-value aggregates between them
        segs = plans[0].segs
        assert sum(s[2] for s in segs if s[0] == "c") >= len(plans[0].calls) - 8 and sum(1 for s in segs if s[0] == "c") <= 8
    assert calls[2] == calls[3] == calls[4] == len(plans[0].calls)          # replays: nothing but the recorded launches
    # another shape: a plan of its own; the first one still replays
    with torch.no_grad():
        y = (torch.randn(2, 3, 128, 128, generator=g) * 50.0).cuda()
        H.LAUNCH_PLANS = False
        wy = tuple(t.clone() for t in m.run_backbone(y))
        H.LAUNCH_PLANS = True
        for _ in range(3):
            gy = m.run_backbone(y)
        assert all(torch.equal(a, b) for a, b in zip(gy, wy))
        again = m.run_backbone(xs[0])
        assert all(torch.equal(a, b) for a, b in zip(again, want[0]))
    assert len(H._LAUNCH_PLANS) == 2
    # with gradients on nothing is planned
    n = len(H._LAUNCH_PLANS)
    m.run_backbone(xs[0])
    assert len(H._LAUNCH_PLANS) == n and all(p.seen in (6, 3) for p in H._LAUNCH_PLANS.values())


def test_only_the_fused_stem_is_planned_and_new_weights_retire_a_plan(det):
    """the un-fused stems rebuild the image with tensor operations a plan never records -- such passes are not
    planned at all; the plan key carries the stem switch and the model's weight epoch; a pass whose recorded launches never saw the
    input pointer is never replayed"""
    import synthetic
    H, m = det
    from maskrcnn_benchmark.modeling.backbone import backbone as B
    g = torch.Generator().manual_seed(3)
    xs = [(torch.randn(2, 3, 128, 160, generator=g) * 50.0).cuda() for _ in range(4)]
    with torch.no_grad():
        H.LAUNCH_PLANS = False
        want = [tuple(t.clone() for t in m.run_backbone(x)) for x in xs]
        H.LAUNCH_PLANS = True
        B._STEM_FUSED[0] = False
        try:
            for x, w in zip(xs, want):      # (the three-launch stem: same bits as the fused one, every input seen afresh)
                got = m.run_backbone(x)
                assert all(torch.equal(a, b) for a, b in zip(got, w))
            assert len(H._LAUNCH_PLANS) == 0
        finally:
            B._STEM_FUSED[0] = True
        odd = (torch.randn(2, 3, 126, 158, generator=g) * 50.0).cuda()   # H, W not multiples of 4: the space-to-depth stem
        a = tuple(t.clone() for t in m.run_backbone(odd))
        for _ in range(3):
            b = m.run_backbone(odd)
        assert len(H._LAUNCH_PLANS) == 0 and all(torch.equal(p, q) for p, q in zip(a, b))
        for x in xs:
            m.run_backbone(x)
        assert len(H._LAUNCH_PLANS) == 1
        # new weights: a new key, the old plan is not replayed (its result would be the old model's)
        shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_shapes.json")))["shapes"]
        m.load_state_dict({k: v.cuda() for k, v in synthetic.make_weights(shapes, seed=5).items()}, strict=False)
        from maskrcnn_benchmark.engine.flat import flatten_model
        flatten_model(m).refresh_planes()
        H.LAUNCH_PLANS = False
        want2 = [tuple(t.clone() for t in m.run_backbone(x)) for x in xs]
        H.LAUNCH_PLANS = True
        assert not torch.equal(want2[0][0], want[0][0])
        for x, w in zip(xs, want2):
            got = m.run_backbone(x)
            assert all(torch.equal(p, q) for p, q in zip(got, w))
        # a pass that does not hand its input to a launch directly: recorded once, found out, never replayed
        seen = []

        def indirect(t):
            seen.append(1)
            z = t.new_zeros((t.shape[0], 4, t.shape[2], t.shape[3]))   # (tensor operations: invisible to a plan)
            z[:, :3] = t
            return (H.maxpool3x3s2(H.nhwc(z)),)

        outs = [H.planned(("indirect",), indirect, x)[0].clone() for x in xs]
        assert len(seen) == len(xs) and not torch.equal(outs[2], outs[3])
        assert len(H._LAUNCH_PLANS) <= H._LP_MAX


def test_dead_plans_give_up_their_memory_pools_at_a_safe_point(det):
    """A plan sits in a reference cycle (its result tensors carry statistics slots whose pool it is) and dies in the cyclic collector,
    which may run at any allocation -- also in the middle of ANOTHER plan's recording, inside torch.cuda.use_mem_pool(), where a
    destroyed MemPool aborts the process in the caching allocator (round 6: once in ten runs of this suite).  LaunchPlan.__del__ hands
    the pool to a graveyard that planned() empties on entry, outside every pool context.  Here: more shapes than the LRU holds, each
    recorded and replayed, with a collector that runs every few allocations; the process must survive, evicted plans must have left
    the table, and their pools must be gone after the next entry."""
    import gc
    H, m = det
    old = gc.get_threshold()
    g = torch.Generator().manual_seed(2)
    shapes = [(2, 3, 128, 160), (2, 3, 160, 128), (1, 3, 192, 128), (2, 3, 128, 128), (1, 3, 128, 192), (1, 3, 160, 160), (2, 3, 192, 192)]
    assert len(shapes) > H._LP_MAX
    gc.set_threshold(20, 2, 2)
    try:
        with torch.no_grad():
            for rnd in range(2):
                for s in shapes:
                    for _ in range(3):   # plain, recorded, replayed
                        x = (torch.randn(*s, generator=g) * 50.0).cuda()
                        got = m.run_backbone(x)
                        H.LAUNCH_PLANS = False
                        want = m.run_backbone(x)
                        H.LAUNCH_PLANS = True
                        for a, b in zip(got, want):
                            assert torch.equal(a, b)
                    del got, want
        gc.collect()
    finally:
        gc.set_threshold(*old)
    assert len(H._LAUNCH_PLANS) <= H._LP_MAX
    with torch.no_grad():
        m.run_backbone((torch.randn(*shapes[0], generator=g) * 50.0).cuda())   # an entry of planned(): the graveyard is emptied
    assert len(H._POOL_GRAVE) == 0
