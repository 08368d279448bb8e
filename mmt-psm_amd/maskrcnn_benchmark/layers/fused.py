"""Autograd nodes of the hot path: every forward AND backward is a sequence of libmmtpsm.so launches.

Gradient convention used by all nodes here (it is what lets ReLU backward ride in GEMM epilogues instead
of costing an elementwise HBM pass per activation):

    the gradient w.r.t. a tensor that is the OUTPUT OF A FUSED ReLU is always delivered ALREADY MASKED
    by (tensor > 0).  Every consumer of such a tensor is one of the nodes below and is told so with
    `input_relu=True`; it applies the mask in the epilogue of its data-gradient GEMM (mask = its saved input).

Weight gradients: fp32-atomic split-K accumulation inside `mmt_conv_wgrad`, straight into the flat gradient
buffer when the model is flattened (engine/flat.py), else returned to autograd as fresh tensors; either way
shared weights (RPN head over 5 levels, box head over sup/unsup passes) sum up.

Reference ops replaced: layers/misc.py:30-64 (Conv2d / ConvTranspose2d), layers/batch_norm.py:19-24,
backbone/resnet.py:254-274 (bottleneck), backbone/fpn.py:43-69, layers/roi_align.py:11-44,
mask_head/loss.py:177-179, detector/generalized_rcnn.py:243-282, box_head/loss.py:185-287.
"""
import torch

from .. import _hip as H


_NONE = {}   # device -> an empty tensor standing for "no downsample weight" among the saved tensors (no fill launch per block)


def _none_like(x):
    t = _NONE.get(x.device)
    if t is None:
        t = _NONE[x.device] = torch.empty((0,), dtype=torch.float32, device=x.device)
    return t


def _dst(t):
    """the parameter's slot in the flat gradient buffer (engine/flat.py), if the model has been flattened"""
    return getattr(t, "_flat_grad", None) if t is not None else None


def _touch(*slots):
    """a backward node wrote into these flat-gradient slots: the parameter 'has a gradient' this step in torch.optim.SGD's
    sense (solver/build.py: FlatSGD leaves parameters without one untouched, like torch's `if p.grad is None: continue`)"""
    for d in slots:
        if d is not None:
            ent = H.GRAD_SLOTS.get(d.data_ptr())
            if ent is not None and ent[0]() is not None:
                ent[0]().touched.add(ent[1])


# Weight gradients on a SIDE stream (MMT_WGRAD_STREAM=1, default): a backward pass is a chain of data gradients with one
# weight gradient hanging off every link -- nothing downstream reads it before the optimiser.  The student's layer3 / layer4
# calls are few-tile launches that leave most of the 256 CUs idle (profiles/r02_conv_table.txt: 40-65 us each against a
# 10 us bound), so the weight gradients run beside the chain instead of inside it, and whoever reads the flat gradient buffer
# (SGD, the all-reduce pieces) joins first (`join_wgrads`).  Only gradients that land in the flat buffer go there (dst_w
# given).  The step is bound by the interpreter time of its launch-issuing threads, so the bookkeeping is batched: jobs are
# collected and handed to the side stream a few at a time behind ONE stream wait (an event record + wait per weight gradient
# cost ~10 us each, 120 times per step), and their operands / split-K workspaces are simply kept alive until the join instead
# of three record_stream calls per job.
import os as _os
_WG_ON = _os.environ.get("MMT_WGRAD_STREAM", "1") != "0"
_WG_BF16 = True   # bf16 configuration: only the DEFERRED (supervised-pass) jobs go to the side stream
_WG_BATCH = int(_os.environ.get("MMT_WG_BATCH", "8"))   # jobs handed over behind one stream wait (round 3, single launches: 1 / 4 / 8 / 16 = 36.8 / 36.6 / 36.4 / 36.4 ms per step; round 6, grouped launches: profiles/r06_history.md)
_WG = {}   # device -> [side stream, launches since the last join, end-of-backward callback queued, pending jobs, kept-alive tensors]


def _wg_stream(dev):
    ent = _WG.get(dev)
    if ent is None:
        # priority -1 like the teacher's stream (engine/MTtrainer.py): HIP deals the streams of a priority class onto a few hardware
        # queues, and once RCCL has created its own a default-priority side stream shares one with the step stream -- under a process
        # group the step cost 37.9 ms with this stream at priority 0 and 36.3 at -1 (world size 1, same box); without one 35.5 either way
        ent = _WG[dev] = [torch.cuda.Stream(device=dev, priority=int(_os.environ.get("MMT_WG_PRIO", "-1"))), 0, False, [], []]
    return ent


def flush_wgrads(ent, dev):
    """hand the collected weight-gradient jobs to the side stream: it waits for everything issued so far on the current
    stream (their operands among it), then takes the launches"""
    jobs = ent[3]
    if not jobs:
        return
    side, keep = ent[0], ent[4]
    gate = _WG_GATE[0]
    if gate is not None:
        ev = gate()
        if ev is None:
            return            # the gate's event does not exist yet: the jobs stay collected (the end of the pass hands them over)
        side.wait_event(ev)
    side.wait_stream(torch.cuda.current_stream(dev))
    # round 6: the batch as grouped launches (include/mmtpsm.h: mmt_conv_wgrad_group) -- the tiles of all its layers fill the chip
    # together, so a layer is cut into fewer pixel ranges than alone; jobs no group takes go out one by one as before
    H.conv_wgrad_group(jobs, side=side, keep=keep)
    for job in jobs:
        keep.append(job[0])    # autograd frees the saved activation / the gradient when the node returns: not before the side
        keep.append(job[1])    # stream has been joined
        if len(job) > 8 and job[8] is not None:
            keep.extend(job[8])
    ent[1] += len(jobs)
    jobs.clear()


# Two passes through the same weights in one step (the labeled and the unlabeled student pass: engine/MTtrainer.py): a layer's two
# weight gradients go out as ONE two-segment launch (mmt_conv_args.x2: half the launches, twice the pixels per block -- the N = 2
# launches were the least efficient group of the step).  Phase "first" (the supervised backward): jobs are parked by the address of
# their gradient slot instead of being handed over; phase "second" (the consistency backward): a job that finds its partner takes
# it along; `finish_wgrad_pairs` hands over whatever found none (a layer only one pass runs through, a skipped consistency branch).
_WG_PAIR = [None]     # None | "first" | "second"
_WG_PARKED = {}       # (address of dw, shapes) -> [parked jobs]


def wgrad_pair_phase(phase):
    _WG_PAIR[0] = phase


def side_stream_for_exchange(dev):
    """the weight-gradient side stream, with every job collected so far handed over and ordered behind the current stream -- what
    a piece of the data-parallel exchange is issued from (engine/MTtrainer.py::BucketedAllReduce._send); None when this
    configuration runs its weight gradients on the step stream"""
    if not _WG_ON or not (H.get_conv_precision() == 3 or _WG_BF16):
        return None
    ent = _wg_stream(dev)
    flush_wgrads(ent, dev)
    ent[0].wait_stream(torch.cuda.current_stream(dev))   # gradients the step stream wrote itself (deconvolution, bias sums)
    ent[1] += 1                                            # (something to join at the end even if no job went over)
    return ent[0]


def release_parked(lo_ptr, hi_ptr):
    """a range of the flat gradient buffer is about to be read (a piece of the data-parallel exchange goes out: everything the
    second pass contributes to it has been issued): parked jobs writing into it will find no partner any more -- issue them now"""
    hit = [k for k in _WG_PARKED if lo_ptr <= k[0] < hi_ptr]
    for k in hit:
        for job in _WG_PARKED.pop(k):
            _wg_stream(job[0].device)[3].append(job)


def finish_wgrad_pairs():
    """after the second pass (or instead of it): the parked jobs that found no partner go to the side stream on their own"""
    _WG_PAIR[0] = None
    if not _WG_PARKED:
        return
    for lst in _WG_PARKED.values():
        for job in lst:
            _wg_stream(job[0].device)[3].append(job)   # (the trainer joins the side stream before the optimiser step)
    _WG_PARKED.clear()
    for dev, ent in _WG.items():
        flush_wgrads(ent, dev)


# Deferral (MMT_WGRAD_DEFER=1, engine/MTtrainer.py): while it is on, jobs are collected but not handed over, and the end-of-backward
# join does nothing -- the supervised pass's weight gradients then go to the side stream in one batch when the trainer says so
# (before it waits for the teacher), behind the whole supervised backward, and run beside the consistency branch, where the GPU
# has room, instead of beside the teacher's backbone, where it has none.
_WG_DEFER = [False]
# Gate (round 5, MMT_WGRAD_GATE=1): instead of holding the supervised pass's jobs back until its backward has been issued, they are
# handed over as they come -- to a side stream that first waits for an EVENT: the end of the teacher's backbone.  From there to the
# teacher's last result the teacher runs latency-bound selection / head kernels and the GPU has room (profiles/r05_phases.txt:
# T.backbone ends at 15.5 ms, the supervised backward at 23.3 ms, and all of its weight gradients used to start only then, beside
# the consistency backward, which is the step's critical chain).  gate() -> the event, or None while it does not exist yet.
_WG_GATE = [None]


def defer_wgrads(on):
    _WG_DEFER[0] = bool(on)


def gate_wgrads(fn):
    _WG_GATE[0] = fn


def flush_deferred_wgrads():
    for dev, ent in _WG.items():
        flush_wgrads(ent, dev)


def _join_wgrads_cb():
    if _WG_GATE[0] is not None:
        for dev, ent in _WG.items():
            ent[2] = False
            flush_wgrads(ent, dev)  # what is left of the pass; NO join: the step stream meets the side stream before the optimiser
        return
    if _WG_DEFER[0]:
        for ent in _WG.values():
            ent[2] = False          # the next backward pass queues its own callback
        return
    join_wgrads()


def join_wgrads(device=None):
    """the current stream waits for every weight gradient issued on the side stream so far (queued as an end-of-backward
    callback by the first side-stream job of a pass, so `.backward()` returns with the gradients ordered on its stream)"""
    for dev, ent in _WG.items():
        ent[2] = False
        if device is None or dev == device:
            flush_wgrads(ent, dev)
            if ent[1]:
                torch.cuda.current_stream(dev).wait_stream(ent[0])
                ent[1] = 0
            ent[4].clear()   # (memory handed back after the wait is re-used behind it in stream order)


def _wgrad(x, g, w, stride, pad, rowscale=None, with_bias=False, dst_w=None, dst_b=None):
    """weight (and bias) gradient.  With flat storage the split-K atomics of `mmt_conv_wgrad` accumulate straight
    into the gradient buffer and None is returned to autograd (no zero-fill, no `grad += dw` pass); otherwise a
    fresh tensor is returned."""
    dw = dst_w if dst_w is not None else torch.zeros_like(H.nhwc(w))
    db = None
    if with_bias:
        db = dst_b if dst_b is not None else torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device)
    if (_WG_ON and dst_w is not None and (dst_b is not None or not with_bias) and not (H.PROFILE is not None and H.PROFILE_ALL)
            and (H.get_conv_precision() == 3 or (_WG_BF16 and _WG_DEFER[0]))):   # (the bf16 configuration is bound by its host threads:
            # 28.5 vs 33.6 ms with the side stream for every job; MMT_WGRAD_BF16=1: only the DEFERRED jobs, handed over while the main thread waits for the teacher)
        ent = _wg_stream(x.device)
        H.wgrad_prepare(x, g)            # reduction passes for operands nobody recorded a maximum of: on THIS stream
        if not ent[2]:
            ent[2] = True
            torch.autograd.Variable._execution_engine.queue_callback(_join_wgrads_cb)
        job = (x, g, tuple(w.shape), stride, pad, dw, rowscale, db)
        phase = _WG_PAIR[0]
        if phase == "first" and H.F16X2 and H.get_conv_precision() == 3:   # (the two-segment launch exists on the fp16 split only)
            _WG_PARKED.setdefault((dw.data_ptr(), x.shape, g.shape, stride, pad), []).append(job)
        else:
            if phase == "second":
                lst = _WG_PARKED.get((dw.data_ptr(), x.shape, g.shape, stride, pad))
                if lst and H.wgrad_pair_ok(lst[0][0], lst[0][1], x, g):
                    first = lst.pop(0)
                    job = first + ((x, g),)     # (the parked pass first: its operands are segment one)
            ent[3].append(job)
            if len(ent[3]) >= _WG_BATCH and not _WG_DEFER[0]:
                flush_wgrads(ent, x.device)
    else:
        H.conv_wgrad(x, g, tuple(w.shape), stride, pad, dw, rowscale, db)
    _touch(dst_w, dst_b if with_bias else None)
    return (None if dst_w is not None else dw), (None if (dst_b is not None or not with_bias) else db)


def _dgrad(g, w, x_shape, stride, pad, scale=None, mask=None, mask_scale=1.0, res=None, res_mode=0, want_planes=False,
           out_dtype=None, rb_site=None, w_owner=None):
    """data gradient of y = conv(x, w) (* scale[co]) w.r.t. x, with optional fused (x>0) mask / residual add;
    out_dtype: the storage type of x (autograd wants the gradient in the tensor's own type: bf16 storage);
    rb_site: the result feeds a plane-fed launch -- this launch's epilogue writes its row-blocked fp16 planes (_hip._rb_produce)"""
    kh = w.shape[2]
    if stride != 1 and kh != 1:
        raise RuntimeError("strided data-gradient is implemented for 1x1 convolutions (STRIDE_IN_1X1) only")
    # the transposed / tap-flipped / BN-scaled weights: straight into packed bf16 planes when the DMA-fed kernels take the
    # data gradient (one launch, nothing materialised in fp32), else as an fp32 tensor
    planes = H.pack_weight_flipped(w, scale)
    kw = dict(w_shape=(w.shape[1], w.shape[0], kh, w.shape[3]), planes=planes, f16_src=(w, scale)) if planes is not None else {}
    wd = None if planes is not None else H.weight_flip_transpose(w, scale, owner=w_owner)
    if stride == 1:
        return H.conv_forward(g, wd, stride=1, pad=kh - 1 - pad, mask=mask, mask_scale=mask_scale, res=res,
                              res_mode=res_mode, want_planes=want_planes, out_dtype=out_dtype, rb_site=rb_site, **kw)
    return H.conv_forward(g, wd, out_stride=stride, out_hw=tuple(x_shape[2:]), mask=mask, mask_scale=mask_scale,
                          res=res, res_mode=res_mode, out_dtype=out_dtype, **kw)


class ConvFn(torch.autograd.Function):
    """y = relu?(conv(x, w) + b)"""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu, input_relu, out_rb=False, din_rb=False):
        # out_rb: y feeds a plane-fed 3x3 launch; din_rb: the gradient w.r.t. x feeds one (the data gradient of the 3x3 layer that
        # produced x) -- this node's launches then write the row-blocked planes from their epilogues (round 6, _hip._rb_produce)
        ctx.acc = getattr(x, "_mmt_acc", None)   # x is an alias of a fork(): its gradient accumulates (_ForkAcc)
        x = H.nhwc(x)
        y = H.conv_forward(x, w, None, b, stride, pad, relu=relu, rb_site=("y", w.data_ptr()) if out_rb else None)
        ctx.save_for_backward(x, w)
        ctx.cfgv = (stride, pad, input_relu, b is not None, din_rb)
        ctx.dst = (_dst(w), _dst(b))
        ctx.w_owner = w   # (the object's identity: _hip.weight_flip_transpose keeps the flipped weights between launches with the same one)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, pad, input_relu, has_b, din_rb = ctx.cfgv
        g = H.nhwc(g)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # first: on the fp16 split it records max |g|, which the weight gradient then reuses
            h = ctx.acc if (stride == 1 and not input_relu and x.dtype == torch.float32) else None
            if h is not None:
                # what the other consumers of x have accumulated so far rides in as this launch's residual operand; the result is
                # the sum, with its statistics (and, where the fork's consumer is plane-fed, its planes) from this epilogue
                dx = _dgrad(g, w, x.shape, stride, pad, None, None, out_dtype=x.dtype, res=h.buf, res_mode=1 if h.buf is not None else 0,
                            rb_site=h.rb_site if h.rb_site is not None else (("dx", w.data_ptr()) if din_rb else None), w_owner=ctx.w_owner)
                if not h.put(dx, True):
                    dx_ret = None
                else:
                    dx_ret = dx
            else:
                dx = _dgrad(g, w, x.shape, stride, pad, None, x if input_relu else None, out_dtype=x.dtype,
                            rb_site=("dx", w.data_ptr()) if din_rb else None, w_owner=ctx.w_owner)
                dx_ret = dx
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw, db = _wgrad(x, g, w, stride, pad, None, has_b, *ctx.dst)
        return (dx_ret if ctx.needs_input_grad[0] else None), dw, db, None, None, None, None, None, None


def conv(x, w, b=None, stride=1, pad=0, relu=False, input_relu=False, out_rb=False, din_rb=False):
    return ConvFn.apply(x, w, b, stride, pad, relu, input_relu, out_rb, din_rb)


def _carry_stats(src, view):
    """a reshaped view of a tensor keeps the statistics slot its producer recorded (same values, shared version counter)"""
    if view is not src:
        am = getattr(src, "_mmt_amax", None)
        if am is not None and am[1] == src._version:
            view._mmt_amax = (am[0], view._version)


class LinearFn(torch.autograd.Function):
    """y = relu?(x @ w.T + b) * mul  -- a 1x1 'conv' over R rows; `mul` carries the scaled dropout mask"""

    @staticmethod
    def forward(ctx, x, w, b, relu, input_relu, in_mask_scale, mul):
        R, K = x.shape
        O = w.shape[0]
        x4 = x.contiguous().view(R, K, 1, 1)
        _carry_stats(x, x4)
        w4 = w.view(O, K, 1, 1)   # (a fresh view per call: _hip's launch plans are tied to the view's BASE object)
        y = H.conv_forward(x4, w4, None, b, relu=relu, mul=None if mul is None else mul.view(R, O, 1, 1))
        ctx.save_for_backward(x4, w4)
        ctx.cfgv = (input_relu, in_mask_scale, b is not None)
        dw_ = _dst(w)
        ctx.dst = (dw_.view(O, K, 1, 1) if dw_ is not None else None, _dst(b))
        out = y.view(R, O)
        _carry_stats(y, out)   # the next Linear reads its input's recorded maximum from the 2-D view
        return out

    @staticmethod
    def backward(ctx, g):
        x4, w4 = ctx.saved_tensors
        input_relu, in_mask_scale, has_b = ctx.cfgv
        R, K = x4.shape[:2]
        O = w4.shape[0]
        gc = g.contiguous()
        g4 = gc.view(R, O, 1, 1)
        _carry_stats(g, gc)       # the statistics the producing launch recorded survive the reshapes (no reduction pass per fc layer)
        _carry_stats(gc, g4)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dw, db = _wgrad(x4, g4, w4, 1, 0, None, has_b, *ctx.dst)
            dw = dw.view(O, K) if dw is not None else None
        if ctx.needs_input_grad[0]:
            d4 = _dgrad(g4, w4, x4.shape, 1, 0, None, x4 if input_relu else None, in_mask_scale)
            dx = d4.view(R, K)
            _carry_stats(d4, dx)
        return dx, dw, db, None, None, None, None


def linear(x, w, b=None, relu=False, input_relu=False, in_mask_scale=1.0, mul=None):
    return LinearFn.apply(x, w, b, relu, input_relu, in_mask_scale, mul)


def batch_slice(t, lo, hi):
    """images lo..hi of an NHWC-dense (N,C,H,W) tensor: a dense view; the bf16 planes a producer attached go along"""
    v = t[lo:hi]
    pl = H.planes_of(t)
    if pl is not None:
        per = t.numel() // t.shape[0]
        v._mmt_planes = (pl[:, lo * per:hi * per], v._version)
    am = getattr(t, "_mmt_amax", None)
    if am is not None and am[1] == t._version:
        v._mmt_amax = (am[0], v._version)   # max over the whole batch: an upper bound for the slice (fp16 split scale)
    rb = getattr(t, "_mmt_rb", None)      # row-blocked fp16 planes [N H][C / 16][W][16]: image-major, so a batch slice is a slice
    if rb is not None and rb[2] == t._version and (len(rb) < 5 or rb[4] is None or hi <= rb[4]):
        per = t.numel() // t.shape[0]   # (planes that cover only the leading images of the batch: slices inside them)
        v._mmt_rb = (rb[0][:, lo * per:hi * per], rb[1], v._version) + tuple(rb[3:4]) + ((None,) if len(rb) > 4 else ())
    return v


# Round 6, second stage (MMT_RB_WIDE=1, off unless measured faster: profiles/r06_history.md): planes ALSO for the operands of the 1x1
# layers' weight gradients and long-K forward / data-gradient launches -- the block outputs and conv2 outputs of a pass that will be
# back-propagated, the gradients that flow between blocks -- so that those launches run plane-fed (wgrad_pl_kernel, conv_pg_kernel).
RB_WIDE = _os.environ.get("MMT_RB_WIDE", "0") != "0"
_PAIR_FWD = [False]   # inside backbone.forward_pair's no-grad forward (its results are back-propagated through `pre=` nodes)


def _wide():
    return RB_WIDE and (torch.is_grad_enabled() or _PAIR_FWD[0])


def bottleneck_forward(x, w1, w2, w3, wd, bn, stride):
    """the 3-4 launches of a bottleneck -> (o1, o2, out)"""
    s1, b1, s2, b2, s3, b3, sd, bd = bn
    # o1 feeds one 3x3 convolution: where that runs on bf16 planes, conv1's epilogue writes them (no split pass)
    mid = w1.shape[0]
    ho, wo = (x.shape[2] + stride - 1) // stride, (x.shape[3] + stride - 1) // stride
    wp = H.planes_wanted_3x3(x.shape[0], mid, ho, wo, w2.shape[0])
    od = torch.bfloat16 if H.bf16_storage() else None   # bf16 activation storage (mode 1): every tensor of the block
    # (round 6, fp16 split: row-blocked fp16 planes with the site's lagged scale instead -- conv2 runs on the tap-strip / plane-fed
    # kernel from layer2 on, and its weight gradient takes o1's planes too)
    o1 = H.conv_forward(x, w1, s1, b1, stride, 0, relu=True, want_planes=wp, out_dtype=od,
                        rb_site=("o1", w2.data_ptr()) if mid >= 128 else None)
    wide = _wide() and mid >= 128
    o2 = H.conv_forward(o1, w2, s2, b2, 1, 1, relu=True, out_dtype=od, rb_site=("o2", w3.data_ptr()) if wide else None)
    r = x if wd is None else H.conv_forward(x, wd, sd, bd, stride, 0, out_dtype=od)
    out = H.conv_forward(o2, w3, s3, b3, 1, 0, relu=True, res=r, res_mode=1, out_dtype=od,
                         rb_site=("out", w3.data_ptr()) if wide else None)
    return o1, o2, out


class BottleneckFn(torch.autograd.Function):
    """BottleneckWithFixedBatchNorm.forward (backbone/resnet.py:254-274) as 3-4 fused launches:
    conv1x1(s)+BN+ReLU -> conv3x3+BN+ReLU -> conv1x1+BN (+ downsample 1x1(s)+BN) + residual + ReLU.
    `pre` = (o1, o2, out) already computed for this input (backbone.py::forward_pair: one N = 4 forward for the two student
    passes of a step): the node then only records what its backward needs."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, bn, stride, pre=None):
        s1, b1, s2, b2, s3, b3, sd, bd = bn
        x = H.nhwc(x)
        o1, o2, out = pre if pre is not None else bottleneck_forward(x, w1, w2, w3, wd, bn, stride)
        ctx.save_for_backward(x, o1, o2, w1, w2, w3, wd if wd is not None else _none_like(x))
        ctx.bn = (s1, s2, s3, sd)
        ctx.stride = stride
        ctx.has_ds = wd is not None
        ctx.dst = (_dst(w1), _dst(w2), _dst(w3), _dst(wd))
        return out

    @staticmethod
    def backward(ctx, g):
        x, o1, o2, w1, w2, w3, wd = ctx.saved_tensors
        s1, s2, s3, sd = ctx.bn
        stride = ctx.stride
        g = H.nhwc(g)  # masked by (out > 0) by the consumer
        d1, d2, d3, dd = ctx.dst
        dw3, _ = _wgrad(o2, g, w3, 1, 0, s3, dst_w=d3)
        d_o2 = _dgrad(g, w3, o2.shape, 1, 0, s3, mask=o2, out_dtype=o2.dtype,
                      want_planes=H.planes_wanted_3x3(o2.shape[0], o2.shape[1], o2.shape[2], o2.shape[3], w2.shape[1]),
                      rb_site=("d_o2", w2.data_ptr()) if o2.shape[1] >= 128 else None)
        dw2, _ = _wgrad(o1, d_o2, w2, 1, 1, s2, dst_w=d2)
        wide = RB_WIDE and o1.shape[1] >= 128
        d_o1 = _dgrad(d_o2, w2, o1.shape, 1, 1, s2, mask=o1, out_dtype=o1.dtype, rb_site=("d_o1", w1.data_ptr()) if wide else None)
        dw1, _ = _wgrad(x, d_o1, w1, stride, 0, s1, dst_w=d1)
        dwd = dx = None
        if ctx.has_ds:
            dwd, _ = _wgrad(x, g, wd, stride, 0, sd, dst_w=dd)
        if ctx.needs_input_grad[0]:
            if ctx.has_ds:
                t = _dgrad(d_o1, w1, (x.shape[0], x.shape[1], g.shape[2], g.shape[3]), 1, 0, s1, out_dtype=x.dtype)  # compact Ho x Wo
                dx = _dgrad(g, wd, x.shape, stride, 0, sd, mask=x, res=t, res_mode=1, out_dtype=x.dtype) if stride > 1 else \
                    _dgrad(g, wd, x.shape, 1, 0, sd, mask=x, res=t, res_mode=1, out_dtype=x.dtype)
            else:
                dx = _dgrad(d_o1, w1, x.shape, 1, 0, s1, mask=x, res=g, res_mode=1, out_dtype=x.dtype,
                            rb_site=("dx", w1.data_ptr()) if wide else None)
        return dx, dw1, dw2, dw3, dwd, None, None, None


def fpn_forward(cs, wi, bi, wl, bl, out_planes=True):
    """the 8 launches of the FPN -> (inner[4], outs[4])"""
    inner = [None] * 4
    # inner_k feeds the 3x3 output convolution, P_k the 3x3 RPN head convolution: planes from the producing epilogues
    wp = [H.planes_wanted_3x3(c.shape[0], wi[k].shape[0], c.shape[2], c.shape[3], wl[k].shape[0]) for k, c in enumerate(cs)]
    od = torch.bfloat16 if H.bf16_storage() else None   # bf16 activation storage: laterals and pyramid levels too
    # (round 6, fp16 split: the same idea with row-blocked fp16 planes and the site's lagged scale, _hip._rb_produce)
    inner[3] = H.conv_forward(cs[3], wi[3], None, bi[3], want_planes=wp[3], out_dtype=od, rb_site=("inner", wl[3].data_ptr()))
    for k in (2, 1, 0):
        inner[k] = H.conv_forward(cs[k], wi[k], None, bi[k], res=inner[k + 1], res_mode=2, want_planes=wp[k], out_dtype=od,
                                  rb_site=("inner", wl[k].data_ptr()))
    # out_planes: True -- every image's P_k feeds the RPN head (student); an int n -- the first n images' do (the teacher's view 0);
    # False -- none
    if out_planes is True:
        site = lambda k: ("P", wl[k].data_ptr())   # noqa: E731
    elif out_planes:
        site = lambda k: ("P", wl[k].data_ptr(), H.RbLead(out_planes))   # noqa: E731
    else:
        site = lambda k: None   # noqa: E731
    outs = [H.conv_forward(inner[k], wl[k], None, bl[k], 1, 1, want_planes=wp[k] and out_planes is True, out_dtype=od,
                           rb_site=site(k)) for k in range(4)]
    return inner, outs


class FPNFn(torch.autograd.Function):
    """FPN.forward (backbone/fpn.py:43-69): 4 lateral 1x1 (+bias) with the nearest-x2 top-down add fused in the
    epilogue, 4 output 3x3 (+bias).  Returns P2..P5 (P6 = P5[::2, ::2] is taken by the caller)."""

    @staticmethod
    def forward(ctx, c2, c3, c4, c5, wi1, bi1, wi2, bi2, wi3, bi3, wi4, bi4, wl1, bl1, wl2, bl2, wl3, bl3, wl4, bl4,
                out_planes=True, pre=None):
        cs = [H.nhwc(c) for c in (c2, c3, c4, c5)]
        wi, bi = (wi1, wi2, wi3, wi4), (bi1, bi2, bi3, bi4)
        wl, bl = (wl1, wl2, wl3, wl4), (bl1, bl2, bl3, bl4)
        inner, outs = pre if pre is not None else fpn_forward(cs, wi, bi, wl, bl, out_planes)
        ctx.save_for_backward(*cs, *inner, *wi, *wl)
        ctx.dst = ([(_dst(wi[k]), _dst(bi[k])) for k in range(4)], [(_dst(wl[k]), _dst(bl[k])) for k in range(4)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g2, g3, g4, g5):
        sv = ctx.saved_tensors
        cs, inner, wi, wl = sv[0:4], sv[4:8], sv[8:12], sv[12:16]
        gs = [g2, g3, g4, g5]
        gs = [H.nhwc(g) if g is not None else torch.zeros_like(inner[k]) for k, g in enumerate(gs)]
        d_in = [None] * 4
        dwl, dbl, dwi, dbi, dcs = [None] * 4, [None] * 4, [None] * 4, [None] * 4, [None] * 4
        for k in range(4):  # finest first: d_inner_k = dgrad(layer_k) + 2x2-sum(d_inner_{k-1})
            d_in[k] = _dgrad(gs[k], wl[k], inner[k].shape, 1, 1, None, res=d_in[k - 1] if k > 0 else None,
                             res_mode=3 if k > 0 else 0, rb_site=("d_in", wi[k].data_ptr()) if RB_WIDE else None)
            dwl[k], dbl[k] = _wgrad(inner[k], gs[k], wl[k], 1, 1, None, True, *ctx.dst[1][k])
        for k in range(4):
            dwi[k], dbi[k] = _wgrad(cs[k], d_in[k], wi[k], 1, 0, None, True, *ctx.dst[0][k])
            if ctx.needs_input_grad[k]:
                dcs[k] = _dgrad(d_in[k], wi[k], cs[k].shape, 1, 0, None, mask=cs[k], out_dtype=cs[k].dtype,  # C_k is a ReLU output
                                rb_site=("dC", wi[k].data_ptr()) if RB_WIDE and k == 3 else None)
        out = list(dcs)
        for k in range(4):
            out += [dwi[k], dbi[k]]
        for k in range(4):
            out += [dwl[k], dbl[k]]
        return tuple(out) + (None, None)


class DeconvFn(torch.autograd.Function):
    """relu?(ConvTranspose2d(k=2, s=2)(x) + b)  (mask_head/roi_mask_predictors.py:34-36): four tap GEMMs that
    scatter straight into the 2x-upsampled NHWC output; w is (Cin, Cout, 2, 2) in channels_last memory, i.e.
    [ci][kh][kw][co] -- already the layout the data-gradient (a stride-2 2x2 conv) wants."""

    @staticmethod
    def forward(ctx, x, w, b, relu, input_relu):
        x = H.nhwc(x)
        w_in = w
        w = H.nhwc(w)
        P, Cin, h, wd_ = x.shape
        Cout = w.shape[1]
        y = H.empty_nhwc(P, Cout, 2 * h, 2 * wd_, x.device)
        wf = w.permute(2, 3, 1, 0).contiguous()  # [kh][kw][co][ci]
        for kh in range(2):
            for kw in range(2):
                H.conv_forward(x, wf[kh, kw].view(Cout, Cin, 1, 1), None, b, relu=relu, out_stride=2,
                               out_hw=(2 * h, 2 * wd_), y_out=y, y_offset=(kh * 2 * wd_ + kw) * Cout)
        ctx.save_for_backward(x, w)
        ctx.cfgv = (input_relu, b is not None)
        ctx.dst = (_dst(w_in), _dst(b))
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        input_relu, has_b = ctx.cfgv
        g = H.nhwc(g)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dst_w, dst_b = ctx.dst
            dw = dst_w if dst_w is not None else torch.zeros_like(w)
            H.conv_wgrad(g, x, tuple(w.shape), 2, 0, dw)  # 'input' = g (28x28), 'dy' = x (14x14)
            _touch(dst_w, dst_b if has_b else None)
            if has_b:
                db = dst_b if dst_b is not None else torch.zeros((w.shape[1],), dtype=torch.float32, device=w.device)
                H.colsum(g, db)
                db = None if dst_b is not None else db
            dw = None if dst_w is not None else dw
        if ctx.needs_input_grad[0]:
            dx = H.conv_forward(g, w, stride=2, pad=0, mask=x if input_relu else None, rb_site=("dx", w.data_ptr()))
        return dx, dw, db, None, None


class RoiAlignFpnFn(torch.autograd.Function):
    """Pooler.forward (modeling/poolers.py:91-121) + _ROIAlign (layers/roi_align.py:11-44) for all levels at once"""

    @staticmethod
    def forward(ctx, rois, levels, res, scales, sr, *feats):
        feats = [H.nhwc(f) for f in feats]
        out = H.roi_align_forward(feats, scales, rois, levels, res, res, sr)
        if out.dtype == torch.float32 and H.F16X2 and H.get_conv_precision() == 3:
            H.stats_of_convex_combination(out, feats)   # fc6 / the mask head scale the pooled tensor from the levels' maxima
        ctx.save_for_backward(rois, levels)
        ctx.cfgv = (res, scales, sr, [tuple(f.shape) for f in feats])
        ctx.fdt = feats[0].dtype
        ctx.acc = [getattr(f, "_mmt_acc", None) for f in feats]   # levels that are aliases of a fork(): their gradient accumulates (_ForkAcc)
        return out

    @staticmethod
    def backward(ctx, g):
        rois, levels = ctx.saved_tensors
        res, scales, sr, shapes = ctx.cfgv
        acc = ctx.acc if ctx.fdt == torch.float32 else [None] * len(shapes)
        into = [h.buf if (h is not None and ctx.needs_input_grad[5 + i]) else None for i, h in enumerate(acc)]
        grads = H.roi_align_backward(g, shapes, scales, rois, levels, res, res, sr, into=into)   # fp32 atomics
        if ctx.fdt != torch.float32:
            grads = [gr.to(ctx.fdt) for gr in grads]   # bf16 storage: the gradient in the level's own type
        out = []
        for i, gr in enumerate(grads):
            if not ctx.needs_input_grad[5 + i]:
                out.append(None)
            elif acc[i] is None:
                out.append(gr)
            else:   # the first contribution travels through autograd, the later ones are in it already
                out.append(gr if acc[i].put(gr, False) else None)
        return (None, None, None, None, None) + tuple(out)


class MaskBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, targets):
        loss, grad = H.mask_bce(logits, labels, targets, 1.0)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return grad * g, None, None


class RPNLossFn(torch.autograd.Function):
    """(objectness loss, box loss) of rpn/loss.py:183-194 in two launches, gradients kept from the forward pass"""

    @staticmethod
    def forward(ctx, obj, reg, labels, regt, pos, neg, beta):
        out, dobj, dreg = H.rpn_loss(obj, reg, labels, regt, pos, neg, beta)
        ctx.save_for_backward(dobj, dreg)
        ctx.shapes = (obj.shape, reg.shape)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g0, g1):
        dobj, dreg = ctx.saved_tensors
        so, sr = ctx.shapes
        return (dobj * g0).view(so), (dreg * g1).view(sr), None, None, None, None, None


class BoxLossFn(torch.autograd.Function):
    """(classification loss, box loss) of box_head/loss.py:118-162 in one launch, gradients kept from the forward pass"""

    @staticmethod
    def forward(ctx, logits, breg, labels, regt, n_rows=None):
        out, dl, db = H.box_loss(logits, breg, labels, regt, n_rows)
        ctx.save_for_backward(dl, db)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g0, g1):
        dl, db = ctx.saved_tensors
        return dl * g0, db * g1, None, None, None


class PSMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, roww, norm, temp, sharpen, kind):
        rl, rg = H.psm_rows(teacher, student, roww, temp, sharpen, kind)
        ctx.save_for_backward(rg * norm)
        return rl.sum() * norm

    @staticmethod
    def backward(ctx, g):
        rg, = ctx.saved_tensors
        return rg * g, None, None, None, None, None, None


class MGDLossFn(torch.autograd.Function):
    """fg_hint_loss (detector/generalized_rcnn.py:243-282): mean over (teacher pyramid x level) masked-L2 terms"""

    @staticmethod
    def forward(ctx, seg, flips, n_levels, *embs):
        students = [H.nhwc(e) for e in embs[:n_levels]]
        nt = (len(embs) - n_levels) // n_levels
        teachers = [[H.nhwc(embs[n_levels + i * n_levels + l]) for i in range(nt)] for l in range(n_levels)]
        C = students[0].shape[1]
        acc = torch.zeros((n_levels, nt + 1), dtype=torch.float32, device=students[0].device)   # one fill for all levels
        masks = []
        for l, s in enumerate(students):
            if s.shape[1] != C:
                raise RuntimeError("MGD: the embeddings of all levels have the same width")
            m = H.mask_pool(seg, s.shape[2], s.shape[3])
            H.mgd_level_forward(s, teachers[l], flips, m, acc[l])
            masks.append(m)
        den = acc[:, nt] * C + 1e-7                       # per level; the same expressions as the per-level form, batched
        ctx.students, ctx.teachers, ctx.masks, ctx.den, ctx.flips = students, teachers, masks, den, flips
        ctx.n_terms = nt * n_levels
        # reference order: for teacher: for level -> mean is order independent up to rounding
        return (acc[:, :nt] / den[:, None]).reshape(-1).mean()

    @staticmethod
    def backward(ctx, g):
        nt = len(ctx.teachers[0])
        coef = (g / ctx.n_terms / ctx.den)[:, None].expand(-1, nt).contiguous()   # [level][teacher]
        grads = [H.mgd_level_backward(s, ts, ctx.flips, m, coef[l])
                 for l, (s, ts, m) in enumerate(zip(ctx.students, ctx.teachers, ctx.masks))]
        return (None, None, None) + tuple(grads) + (None,) * (len(ctx.teachers[0]) * len(ctx.students))


class ReluGradMaskFn(torch.autograd.Function):
    """identity whose backward applies (x > 0): the adapter between a fused-ReLU output and consumers that are
    NOT nodes of this file (index / cat / library ops), so that the producer still receives a pre-masked gradient"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return g * (x > 0).to(g.dtype)


def relu_grad_mask(x):
    return ReluGradMaskFn.apply(x) if x.requires_grad else x


class SplitBatchFn(torch.autograd.Function):
    """x[:n], x[n:] along the batch dimension.  Plain slicing would make autograd materialise two zero-filled
    full-size gradients and add them; here the backward is one concatenation (a single pass over the gradient)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        n, shape = ctx.n, ctx.shape
        out = H.empty_nhwc(shape[0], shape[1], shape[2], shape[3], (ga if ga is not None else gb).device)
        if ga is not None:
            out[:n].copy_(ga)
        else:
            out[:n].zero_()
        if gb is not None:
            out[n:].copy_(gb)
        else:
            out[n:].zero_()
        return out, None


def split_batch(x, n):
    return SplitBatchFn.apply(x, n)


_FORK_ON = True


_FORK_ACC = _os.environ.get("MMT_FORK_ACC", "1") != "0"


class _ForkAcc(object):
    """the gradient of a forked tensor, ACCUMULATED by the consumers that know how instead of written once per consumer and summed
    (round 6): a ROIAlign backward adds its atomics to what is there, a convolution's data gradient takes what is there as its
    residual operand.  buf: the sum so far (None: nobody yet); fresh: buf carries the statistics / planes of its current values (a
    convolution wrote it last); seen: every tensor that ever was `buf` (autograd still hands the first one to ForkFn.backward)"""
    __slots__ = ("buf", "fresh", "seen", "rb_site")

    def __init__(self, rb_site):
        self.buf, self.fresh, self.seen, self.rb_site = None, False, [], rb_site

    def put(self, t, fresh):
        first = self.buf is None
        if t is not self.buf:
            self.seen.append(t)
        self.buf, self.fresh = t, fresh
        if not fresh:   # changed in place through raw pointers: what its producer recorded about it no longer holds
            for a in ("_mmt_amax", "_mmt_rb", "_mmt_planes"):
                if getattr(t, a, None) is not None:
                    setattr(t, a, None)
        return first


class ForkFn(torch.autograd.Function):
    """x -> n aliases of x, one per consumer.  Autograd would add the consumers' gradients pairwise with library launches
    (n - 1 of them) and the first fp16-split consumer of the sum would then take a reduction pass for its scale; here the
    backward is ONE `mmt_sum_stats` launch that adds them and records max / mean |.| of the sum on the way."""

    @staticmethod
    def forward(ctx, x, n, rb_site=None, acc=None):
        ctx.set_materialize_grads(False)   # an alias nobody back-propagated through arrives as None, not as a zero-filled tensor
        ctx.rb_site = rb_site
        ctx.acc = acc
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        h = ctx.acc
        if h is not None and h.buf is not None:
            # the consumers that accumulate handed over ONE tensor between them (the first of them) and None since: what counts is
            # the accumulator as it stands now; consumers that do not know the protocol sent tensors of their own
            gs = [g for g in gs if not any(g is b for b in h.seen)]
            if not gs and h.fresh:
                return h.buf, None, None, None   # a convolution wrote it last: statistics and planes are those of the sum
            gs = [h.buf] + gs
        if not gs:
            return None, None, None, None
        if len(gs) == 1:
            return gs[0], None, None, None
        if gs[0].dim() == 4:
            gs = [H.nhwc(g) for g in gs]
        else:
            gs = [g.contiguous() for g in gs]
        while len(gs) > 4:
            gs = [H.sum_stats(gs[:4])] + gs[4:]
        return H.sum_stats(gs, ctx.rb_site), None, None, None


def fork(x, n, rb_site=None):
    """n handles of x for n consumers whose gradients are summed in one launch (the fp16-split arithmetic on fp32 tensors only:
    elsewhere autograd's own accumulation stays).  rb_site: the summed gradient feeds a plane-fed data-gradient launch -- the sum
    launch leaves its row-blocked planes (_hip.sum_stats)"""
    if n < 2:
        return (x,) * n
    if not (_FORK_ON and x.requires_grad and x.is_cuda and x.dtype == torch.float32 and H.F16X2 and H.get_conv_precision() == 3):
        return (x,) * n
    acc = _ForkAcc(rb_site) if (_FORK_ACC and x.dim() == 4) else None
    outs = ForkFn.apply(x, n, rb_site, acc)
    if acc is not None:
        for o in outs:
            o._mmt_acc = acc
    rb = getattr(x, "_mmt_rb", None)
    if rb is not None and rb[2] == x._version:
        for o in outs:
            o._mmt_rb = (rb[0], rb[1], o._version) + tuple(rb[3:])
    for a in ("_mmt_planes", "_mmt_amax"):   # planes / statistics of the tensor go along with its aliases
        v = getattr(x, a, None)
        if v is not None and v[1] == x._version:
            for o in outs:
                setattr(o, a, (v[0], o._version))
    return outs


def fork_levels(levels, n):
    """fork() of every tensor of a pyramid -> n pyramids (the summed gradient of a level feeds the data gradient of the FPN's 3x3
    output convolution: planes from the sum launch)"""
    cols = [fork(t, n, ("gP", k)) for k, t in enumerate(levels)]
    return [tuple(c[i] for c in cols) for i in range(n)]


class RelationAttentionFn(torch.autograd.Function):
    """IR-Net's multi-head geometric relation attention (reference relation/relation_module.py:33-90) as one launch each way
    (csrc/relation.hip): q, k (C*N, G*DQ), wg (C*N*N, G) = the fused-ReLU output of WG, v (C*N, G*DV), bias (G*DV)
    -> (N, C, G*DV).  The gradient w.r.t. wg comes out masked by the clamp (wg >= 1e-6), i.e. already in the convention of a
    fused-ReLU output (see the top of this file)."""

    @staticmethod
    def forward(ctx, q, k, wg, v, bias, C, N, G, topk, scale):
        ctx.types = tuple(t.dtype for t in (q, k, wg, v))
        q, k, wg, v = (t.float().contiguous() for t in (q, k, wg, v))   # (fp32 in every shipped configuration: no copies)
        out, P = H.relation_attention_fwd(q, k, wg, v, bias.float(), C, N, G, topk, scale)
        ctx.save_for_backward(q, k, wg, v, P)
        ctx.cfgv = (C, N, G, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, wg, v, P = ctx.saved_tensors
        C, N, G, scale = ctx.cfgv
        g = g.contiguous()
        g = g.float().contiguous()
        grads = H.relation_attention_bwd(q, k, wg, v, P, g, C, N, G, scale)
        dq, dk, dwg, dv = (t if t.dtype == d else t.to(d) for t, d in zip(grads, ctx.types))
        db = g.sum((0, 1)) if ctx.needs_input_grad[4] else None
        return dq, dk, dwg, dv, db, None, None, None, None, None


class CIAMFn(torch.autograd.Function):
    """IR-Net's cross-instance attention (reference relation/mask_relation_module.py:199-242) over all (image, class) groups
    of the batch: one forward launch, two backward launches (csrc/relation.hip): x (n, C, H, W), group (n,) int64 with
    equal ids contiguous, gamma (1,) -> gamma * attention(x) + x"""

    @staticmethod
    def forward(ctx, x, group, gamma):
        ctx.xtype = x.dtype
        x = x.float().contiguous()          # NCHW-dense rows [C][H W] per instance
        out, A, J = H.ciam_fwd(x, group.contiguous(), gamma)
        ctx.save_for_backward(x, group, gamma, A, J)
        return out if out.dtype == ctx.xtype else out.to(ctx.xtype)

    @staticmethod
    def backward(ctx, g):
        x, group, gamma, A, J = ctx.saved_tensors
        dx, dgamma = H.ciam_bwd(x, group.contiguous(), gamma, A, J, g.float().contiguous())
        return (dx if dx.dtype == ctx.xtype else dx.to(ctx.xtype)), None, dgamma
