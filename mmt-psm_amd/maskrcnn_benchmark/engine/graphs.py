"""hipGraph capture of the backbone passes (ResNet body + FPN) of a step.

The three backbone passes of a mean-teacher step (student on the labeled crops, student on the unlabeled view, teacher on
the K x {plain, mirrored} views) have fixed shapes and no host decisions: ~100 launches forward and ~190 backward each,
every one costing this thread 8-20 us of Python, ctypes and launch work -- 8 ms forward and 6 ms backward per step, on a
thread that the box / mask / RPN glue keeps busy anyway.  Captured once per (pass, shape), a pass is one graph launch.

One `BackboneGraph` = one pass: a forward graph and (training) a backward graph sharing a memory pool, in which the saved
activations live from the forward replay to the backward replay.  The two student passes of a step therefore use two
instances (the supervised backward runs after BOTH forwards).  To autograd a pass is ONE node (`_GraphFn`): its backward
copies the pyramid gradients into static buffers and replays; the weight gradients accumulate into the flat gradient
buffer from inside the graph exactly as they do eagerly (layers/fused.py), so nothing is returned for them.

What a replay does NOT do is run Python: the `touched` bookkeeping of engine/flat.py is recorded at capture and re-applied
per replay, and the stage hooks of the bucketed all-reduce never fire -- the caller falls back to the eager pass when those
hooks are installed (world size > 1 with MMT_BUCKETED_ALLREDUCE).  Captures are made on the calling thread with no other
thread launching (MTtrainer captures all three before it starts the teacher thread).

Measured on 1 x MI355X (tools/graph_replay_cost.py, bench.py), ROCm 7.2: a replayed pass takes the device time of the eager
pass (13.09 vs 13.04 ms teacher forward, 9.81 vs 9.75 ms student forward + backward) and 0.05 / 0.26 ms of host time
instead of 1.6 / 4.2 ms.  In the step: fp32-grade arithmetic 45.9 (student graphs) / 45.5 (teacher graph) vs 45.6 ms eager --
the step is bound by the device there; bf16 arithmetic + bf16 storage 27.7 (student graphs) vs 29.0 ms.  BOTH models
graphed while the two streams overlap: 83 ms -- two graphs replaying concurrently from two threads serialise badly in
this runtime.  Hence OFF by default (MMT_GRAPHS=0); MMT_GRAPHS=2 (student passes only) is the useful setting for the
launch-bound bf16 configuration."""
import torch

from .. import _hip as H


class _GraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inst, anchor, x):
        inst.x_s.copy_(x)
        inst.g_fwd.replay()
        ctx.inst = inst
        ctx.set_materialize_grads(False)
        return inst.outputs()

    @staticmethod
    def backward(ctx, *gs):
        inst = ctx.inst
        for sg, g in zip(inst.static_grads, gs):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        inst.g_bwd.replay()
        if inst.flat is not None:
            inst.flat.touched |= inst.touched
        return None, None, None


class BackboneGraph(object):
    def __init__(self, backbone, x, train, flat=None):
        self.flat, self.train = flat, train
        self.stream = torch.cuda.Stream(device=x.device)
        self.x_s = x.detach().clone()
        # the anchor makes the node differentiable (the image itself needs no gradient)
        self.anchor = torch.zeros((), device=x.device, requires_grad=True) if train else None
        before = set(flat.touched) if flat is not None else set()
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            for _ in range(2):  # first-use allocations of the library / the weight-plane registry happen outside the capture
                self._run(backbone, None)
        torch.cuda.current_stream(x.device).wait_stream(self.stream)
        torch.cuda.synchronize(x.device)
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, stream=self.stream, capture_error_mode="thread_local"):
            with torch.set_grad_enabled(train):
                outs = tuple(backbone(self.x_s))
        self.static_outs = outs
        self.planes = [getattr(o, "_mmt_planes", None) for o in outs]
        self.g_bwd = None
        self.static_grads = None
        if train:
            self.static_grads = [torch.zeros_like(o) for o in outs]
            self.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool(), stream=self.stream, capture_error_mode="thread_local"):
                torch.autograd.backward([o for o in outs if o.requires_grad],
                                        [g for o, g in zip(outs, self.static_grads) if o.requires_grad])
        self.touched = (set(flat.touched) - before) if flat is not None else set()
        if flat is not None:   # the warm-up passes added zeros to the gradient buffer; their flags are not this step's
            flat.touched = before
        torch.cuda.synchronize(x.device)

    def _run(self, backbone, _):
        with torch.set_grad_enabled(self.train):
            outs = tuple(backbone(self.x_s))
        if self.train:
            req = [o for o in outs if o.requires_grad]
            torch.autograd.backward(req, [torch.zeros_like(o) for o in req])

    def outputs(self):
        """fresh tensor objects on the static output storage (the captured ones carry the capture-time autograd graph)"""
        res = []
        for o, pl in zip(self.static_outs, self.planes):
            d = o.detach()
            if pl is not None:  # bf16 planes written by the producing epilogue inside the graph: refreshed by every replay
                d._mmt_planes = (pl[0], d._version)
            res.append(d)
        return tuple(res)

    def __call__(self, x):
        if self.train:
            return _GraphFn.apply(self, self.anchor, x)
        self.x_s.copy_(x)
        self.g_fwd.replay()
        return self.outputs()


class BackboneGraphs(object):
    """the captured passes of one model, keyed by (slot, input shape, grad mode, arithmetic mode)"""

    def __init__(self, model, flat=None):
        self.model, self.flat, self.table = model, flat, {}

    def key_of(self, shape, slot):
        return (slot, tuple(shape), bool(torch.is_grad_enabled() and self.model.training), H.get_conv_precision(),
                H.bf16_storage())

    def key(self, x, slot):
        return self.key_of(x.shape, slot)

    def usable(self):
        body = self.model.backbone.body
        # not with the stage hooks of the bucketed all-reduce, not under a test's replay, not while bench.py brackets every
        # launch of the dominant kernel with events (its roofline leg): those need the Python of the eager pass
        return body.grad_ready is None and getattr(self.model, "_replay", None) is None and H.PROFILE is None

    def prepare(self, x, slot):
        k = self.key(x, slot)
        if k not in self.table:
            self.table[k] = BackboneGraph(self.model.backbone, x, k[2], self.flat)
        return self.table[k]

    def __call__(self, x, slot=0):
        if not self.usable():
            return tuple(self.model.backbone(x))
        return self.prepare(x, slot)(x)
