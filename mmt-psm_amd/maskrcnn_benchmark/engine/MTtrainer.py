"""Mean-teacher engine (reference: engine/MTtrainer.py:16-281): per-iteration
  [A] student(image, target)  [B] teacher.forward_teacher(K augs)  [C] student.forward_student(aug, teacher dict)
  [D] weighted loss sum -> backward -> SGD  [E] EMA teacher update.
Arithmetic of loss weighting (weight_sum_losses, including the reference's ramp-down-with-ramp-up-length quirk)
and of the EMA (alpha = min(1 - 1/(it+1), MT.ALPHA)) is reproduced; EMA and SGD are single launches over flat
storage; with WORLD_SIZE > 1 the student gradients are averaged by ONE RCCL all-reduce over the flat gradient
buffer (new functionality: the reference has no gradient synchronisation at all)."""
import logging
import os
import time
from functools import partial

import torch
import torch.distributed as dist

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.engine.flat import flatten_model
from maskrcnn_benchmark.layers import fused
from maskrcnn_benchmark.utils.miscellaneous import sigmoid_rampdown, sigmoid_rampup


# The two student passes' weight gradients of a layer as ONE two-segment launch (mmt_conv_args.x2).  Built, tested
# (tests/test_f16x2_gpu.py::test_wgrad_two_segments, test_train_step_gpu.py::test_paired_weight_gradients_equal_separate) and
# measured: 119 -> 68 weight-gradient launches per step, but the step is SLOWER, 35.7 -> 37.5 ms in three same-box alternations
# (profiles/r04_history.md): the supervised pass's jobs used to run in the ~4 ms between the supervised backward and the
# consistency backward, where the device waits for the teacher and the host; paired, they wait for their partner and all
# weight-gradient work lands beside the consistency backward, the most contended stretch of the step.  Off.
_WGRAD_PAIR = os.environ.get("MMT_WGRAD_PAIR", "0") != "0"
_LOSS_ROOTS = os.environ.get("MMT_LOSS_ROOTS", "1") != "0"   # backward from the losses as roots with the weights as seeds (see _backward_roots)
_WGRAD_GATE = os.environ.get("MMT_WGRAD_GATE", "0") != "0"     # ... or as they come, behind the end of the teacher's backbone (round 5: +0.3 .. +0.8 ms, off)
_WGRAD_DEFER = os.environ.get("MMT_WGRAD_DEFER", "1") != "0"   # supervised weight gradients in one batch after the supervised backward (0: interleaved, the A/B alternative)


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_loss_dict(loss_dict):
    """MTtrainer.py:16-42: rank 0 gets the average (logging only)"""
    ws = get_world_size()
    if ws < 2:
        return dict(loss_dict)
    with torch.no_grad():
        names = sorted(loss_dict.keys())  # same order on every rank
        allv = torch.stack([loss_dict[k].reshape(()) for k in names], 0)
        dist.reduce(allv, dst=0)
        if dist.get_rank() == 0:
            allv /= ws
        return {k: v for k, v in zip(names, allv)}


def mt_weight(step, rampup_length, rampdown_length, total_length, l=1, start_mt=1000):
    if (step - start_mt) < rampup_length and (step - start_mt) > 0:
        return l * sigmoid_rampup(step - start_mt, rampup_length)
    if (total_length - step) < rampdown_length:
        return l * sigmoid_rampdown(total_length - step, rampup_length)  # sic: ramp-UP length (MTtrainer.py:92)
    return l


def weight_sum_losses(loss_dict, step, rampup_length, rampdown_length, total_length, l=1, balanced=None, start_mt=1000):
    """MTtrainer.py:67-109"""
    w = mt_weight(step, rampup_length, rampdown_length, total_length, l, start_mt)
    out = {}
    for k, v in loss_dict.items():
        v = w * v if "mt" in k else v
        if balanced is not None and k in balanced:
            v = v * balanced[k]
        out[k] = v
    return out


def init_teacher_weight(model_s, model_t):
    flatten_model(model_t).data.copy_(flatten_model(model_s).data)
    flatten_model(model_t).refresh_planes()


def allreduce_gradients(flat):
    """data-parallel exchange step: average the flat student gradient over ranks (RCCL over xGMI)"""
    ws = get_world_size()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)
        if ws > 1:
            flat.grad.mul_(1.0 / ws)


_FLAG_SYNC = {}   # process-wide: the flag exchange's own communicator and stream (built once, collectively)


def sync_touched(flat):
    """Which parameters "have a gradient" this step must be the same on every rank: the all-reduce hands every rank the same
    averaged gradient, but `flat.touched` (what FlatSGD.step updates, torch SGD's `p.grad is None: continue`) is filled by
    THIS rank's backward nodes.  A rank whose teacher found no boxes skipped the consistency branch and never touched the
    hint adaptors while the others did -- and a supervised pass can take a degenerate route on one rank only (crops without
    ground truth: no mask-head positives) --; updating parameters on some ranks only would let the students, and with them the
    EMA teachers, drift apart for good.  The update set is therefore the UNION over ranks, EVERY step: one MAX
    all-reduce of a per-parameter flag vector (a few hundred bytes).
    Round 5: the flags are host data (filled while backward was ISSUED), so their exchange does not have to wait for the device:
    it runs on a communicator and a stream of its own -- behind nothing the step has queued -- and the read-back returns as soon
    as those few hundred bytes have crossed, with the backward pass still running (round 4 issued it on the gradient's
    communicator, behind every gradient piece, and read it back with a blocking `.tolist()`: a device drain in front of SGD)."""
    if get_world_size() < 2:
        return
    ent = flat.__dict__.get("_touch_sync")
    if ent is None:
        names = [n for n, (o, _) in sorted(flat.index.items(), key=lambda kv: kv[1][0]) if o < flat.n_trainable]
        host = torch.zeros((len(names),), dtype=torch.int32)
        back = torch.zeros((len(names),), dtype=torch.int32)
        if flat.grad.is_cuda:
            host, back = host.pin_memory(), back.pin_memory()
        ent = flat._touch_sync = (names, host, back)
    names, host, back = ent
    touched = flat.touched
    for i, n in enumerate(names):
        host[i] = 1 if n in touched else 0
    if "group" not in _FLAG_SYNC:   # (collective: every rank reaches its first step's exchange)
        _FLAG_SYNC["group"] = dist.new_group()
        _FLAG_SYNC["stream"] = torch.cuda.Stream(device=flat.grad.device) if flat.grad.is_cuda else None
    side = _FLAG_SYNC["stream"]
    if side is None:
        flags = host.clone()
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=_FLAG_SYNC["group"])
        got = flags.tolist()
    else:
        with torch.cuda.stream(side):
            flags = host.to(flat.grad.device, non_blocking=True)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=_FLAG_SYNC["group"])
            back.copy_(flags, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        done.synchronize()        # the side stream's few operations only
        got = back.tolist()
    flat.touched.update(n for n, f in zip(names, got) if f)


class BucketedAllReduce(object):
    """The same exchange, started while the backward pass is still running.  The flat gradient is cut at the stage
    boundaries of the backbone ([layer2 | layer3 | layer4 | FPN + heads]: the weights region of engine/flat.py is in
    parameter order); backward runs through these pieces from the right, and a tensor hook on each stage output
    (modeling/backbone/backbone.py) says when the pieces to its right are final.  Each finished piece goes out as an
    asynchronous all-reduce that overlaps the backward of the earlier stages; whatever is left (layer2, the bias region)
    is reduced at the end.  One scaling pass by 1/world afterwards, as in `allreduce_gradients`.

    The sequence of collectives is the SAME on every rank whatever happened on it: always the stage pieces in the fixed
    order layer4, layer3, layer2, layer1 (each exactly one all-reduce over the same [lo, hi)), then the fixed remainder
    ranges in ascending order.  A piece whose hook did not fire often enough on this rank -- its teacher found no boxes
    and the consistency branch was skipped, so the second backbone pass was never back-propagated -- is sent from
    finish() as its own collective instead of being merged into a neighbour (ranks that did fire it sent exactly that)."""

    # a hook on the output of stage K fires when everything AFTER stage K is done; "heads" = hooks on the pyramid levels (round 5):
    # the RPN / box / mask heads, adaptors and relation modules -- half of the 82 MB that used to leave with the FPN -- go out when
    # the heads' backward ends, ~10 ms before the FPN's (profiles/r04_bench_rccl_world1.json: that piece was issued 3.5 ms before
    # the end of a 36 ms backward)
    AFTER = {"heads": None, "layer4": "backbone.fpn.", "layer3": "backbone.body.layer4.", "layer2": "backbone.body.layer3.",
             "layer1": "backbone.body.layer2."}
    ORDER = ("heads", "layer4", "layer3", "layer2", "layer1")  # firing order of the hooks in a backward pass

    def __init__(self, flat, body):
        self.flat, self.body = flat, body
        names = [n for n, _ in flat._named if n in flat.index and flat.index[n][0] < flat.n_weights]
        cuts = {}
        for stage, prefix in self.AFTER.items():
            if prefix is None:   # everything behind the backbone, in parameter order
                last_bb = max((i for i, n in enumerate(names) if n.startswith("backbone.")), default=-1)
                first = names[last_bb + 1] if 0 <= last_bb < len(names) - 1 else None
            else:
                first = next((n for n in names if n.startswith(prefix)), None)
            if first is not None:
                cuts[stage] = flat.index[first][0]
        # piece that becomes final when `stage` fires: [cut(stage), cut(previous firing stage) or n_weights)
        self.pieces, hi = {}, flat.n_weights
        for s_ in (s for s in self.ORDER if s in cuts):
            lo = cuts[s_]
            if lo < hi:
                self.pieces[s_] = (lo, hi)
            hi = min(hi, lo)
        # what no stage piece covers (the weights before the first cut, the bias region): fixed ranges, sent last
        self.rest, pos, n = [], 0, flat.grad.numel()
        for lo, hi in sorted(self.pieces.values()) + [(n, n)]:
            if pos < lo:
                self.rest.append((pos, lo))
            pos = max(pos, hi)
        self.reset()

    def reset(self):
        self.registered, self.fired, self.sent, self.works = {}, {}, set(), []
        self._marks = []

    def install(self):
        self.reset()
        self.body.grad_ready = self._event
        # MMT_DIST_TRACE=1: per-piece issue / completion times on the device clock (events on the step stream), so that the
        # first multi-GPU run shows at once whether the exchange hides behind the backward pass (`last_trace`)
        self.tracing = os.environ.get("MMT_DIST_TRACE") == "1" and self.flat.grad.is_cuda
        if self.tracing:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()

    def _send(self, lo, hi):
        """one piece of the flat gradient goes out.  Its last contributions are weight gradients on the SIDE stream (layers/fused.py)
        and whatever the step stream has issued up to this hook.  Round 4: the collective is issued FROM the side stream, behind a
        side-waits-for-step-stream edge -- RCCL's stream then orders itself behind both, and the step stream is never made to wait
        (round 3 joined the side stream into the step stream here: four stalls of the data-gradient chain per backward pass, part of
        the +2.4 ms a process group cost at world size 1)."""
        from maskrcnn_benchmark.layers import fused as F_
        g = self.flat.grad
        F_.release_parked(g.data_ptr() + 4 * lo, g.data_ptr() + 4 * hi)   # supervised-pass jobs still waiting for a partner that will not come
        side = F_.side_stream_for_exchange(g.device) if g.is_cuda else None
        if getattr(self, "tracing", False):   # when backward reached this hook, on the STEP stream's clock (the mark used
            e = torch.cuda.Event(enable_timing=True)   # to be recorded inside the side-stream context and measured that queue)
            e.record()
            self._marks.append([lo, hi, e, None, time.perf_counter()])
        if side is None:
            F_.join_wgrads()   # no side stream in this configuration: the piece is final on the step stream
            self._issue(lo, hi)
            return
        with torch.cuda.stream(side):
            self._issue(lo, hi)

    def _issue(self, lo, hi):
        self.works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def _event(self, stage, what):
        if what == "registered":
            self.registered[stage] = self.registered.get(stage, 0) + 1
            return
        self.fired[stage] = self.fired.get(stage, 0) + 1
        # several backbone passes in one step: the piece is final when the LAST registered pass has fired
        if stage in self.pieces and stage not in self.sent and self.fired[stage] == self.registered.get(stage, 0):
            # earlier pieces of the fixed order that this rank never completed go first, so the order stays the same
            for s_ in self.ORDER:
                if s_ == stage:
                    break
                if s_ in self.pieces and s_ not in self.sent:
                    self._send(*self.pieces[s_])
                    self.sent.add(s_)
            self._send(*self.pieces[stage])
            self.sent.add(stage)

    def finish(self):
        """after backward: reduce what no hook covered (fixed order), wait for everything, scale"""
        self.body.grad_ready = None
        for s_ in self.ORDER:
            if s_ in self.pieces and s_ not in self.sent:
                self._send(*self.pieces[s_])
                self.sent.add(s_)
        tracing = getattr(self, "tracing", False)
        if tracing:
            bw_end = torch.cuda.Event(enable_timing=True)
            bw_end.record()
            n_early = len(self._marks)
        for lo, hi in self.rest:
            self._send(lo, hi)
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
            if tracing:
                e = torch.cuda.Event(enable_timing=True)
                e.record()       # the step stream gets here when piece i has arrived
                self._marks[i][3] = e
        ws = get_world_size()
        if ws > 1:
            self.flat.grad.mul_(1.0 / ws)
        if tracing:
            torch.cuda.synchronize()
            self.last_trace = {
                "backward_end_ms": round(self._t0.elapsed_time(bw_end), 3), "pieces_sent_before_backward_end": n_early,
                # what the exchange costs the step: from the end of backward to the arrival of the last piece (0 = fully hidden)
                "exposed_comm_ms": round(max([self._t0.elapsed_time(m[3]) for m in self._marks] + [0.0]) - self._t0.elapsed_time(bw_end), 3),
                "pieces": [{"range": [lo, hi], "mbytes": round((hi - lo) * 4 / 1e6, 2), "issued_ms": round(self._t0.elapsed_time(a), 3),
                            "arrived_ms": round(self._t0.elapsed_time(b), 3)} for lo, hi, a, b, _ in self._marks]}
            # (round 6) what the same issue schedule would cost at 8 GPUs under the xGMI model -- a prediction, never a measurement
            self.last_trace["predicted_8gpu"] = predict_exposed_comm(self.last_trace["pieces"], self.last_trace["backward_end_ms"], 8)
        self.reset()


XGMI_LINK_GBS = 153.0   # xGMI is a full mesh, 7 links x ~153 GB/s per GPU


def predict_exposed_comm(pieces, backward_end_ms, world=8, link_gbs=XGMI_LINK_GBS, latency_ms=0.05):
    """What the bucketed exchange would cost a step at `world` GPUs, from the issue times a (world-size-1) trace recorded and SURVEY
    section 5's xGMI model -- a number for the first real multi-GPU run to be wrong against (no 8-GPU node was
    ever available to the builder).  Two collective algorithms bracket what RCCL does on a full mesh:
      ring        2 (world - 1) hops of bytes / world over ONE link each:  t = 2 (world - 1) / world * bytes / link
      direct      reduce-scatter + all-gather over all world - 1 links at once:  t = 2 / world * bytes / link
    Pieces go out in issue order on one communicator: piece i starts at max(its issue time, the end of piece i - 1) and takes
    t(bytes) + a fixed launch / synchronisation latency.  exposed = end of the last piece - end of backward (>= 0).
    pieces: [{"mbytes", "issued_ms"}, ...] as in `dist_trace`."""
    out = {}
    for name, factor in (("ring", 2.0 * (world - 1) / world), ("direct", 2.0 / world)):
        t = 0.0
        for p in sorted(pieces, key=lambda q: q["issued_ms"]):
            t = max(t, p["issued_ms"]) + latency_ms + factor * p["mbytes"] * 1e6 / (link_gbs * 1e9) * 1e3
        out[name] = {"last_arrival_ms": round(t, 3), "exposed_comm_ms": round(max(0.0, t - backward_end_ms), 3)}
    out["model"] = ("SURVEY section 5: %d GPUs, full-mesh xGMI at %.0f GB/s per link, %.2f ms fixed cost per collective; ring = "
                    "2 (N - 1) / N x bytes over one link, direct = 2 / N x bytes over all links; issue times from this run's trace"
                    % (world, link_gbs, latency_ms))
    return out


def teacher_checksum(flat):
    """exact (integer) checksum of a flat parameter buffer: the sum of its words read as int32, in int64"""
    return flat.data.view(torch.int32).to(torch.int64).sum()


def check_teacher_identity(flat_t):
    """the teachers are never exchanged -- they stay identical on all ranks because every rank applies the
    same EMA to the same (all-reduced) student from the same initial weights.  Asserted here with ONE small collective:
    max over ranks of (checksum, -checksum); equal teachers <=> max(c) == -max(-c)."""
    if get_world_size() < 2:
        return True
    c = teacher_checksum(flat_t)
    v = torch.stack([c, -c])
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    hi, neg_lo = v.tolist()
    if hi != -neg_lo:
        raise RuntimeError("teacher weights diverged between ranks (checksum max %d != min %d)" % (hi, -neg_lo))
    return True


class MTtrainer(object):
    def __init__(self, model_s, model_t, data_loader, optimizer, scheduler, ckpt_s, ckpt_t, checkpoint_period, cfg):
        self.cfg = cfg
        self.logger = logging.getLogger("maskrcnn_benchmark.trainer")
        self.scheduler, self.optimizer = scheduler, optimizer
        self.max_iter = len(data_loader["source"])
        self.start_iter = 0
        self.student, self.teacher = model_s, model_t
        self.student_bs, self.teacher_bs = cfg.MT.AUG_S, cfg.MT.AUG_K
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.checkpoint_period, self.ckpt_s, self.ckpt_t = checkpoint_period, ckpt_s, ckpt_t
        self.lambda_value, self.alpha = cfg.MT.LAMBDA, cfg.MT.ALPHA
        self.start_mt = cfg.MT.START_MT
        self.balanced_weight = {"mt_classifier": cfg.MT.CLS_LOSS, "nms_loss": cfg.MODEL.RELATION_NMS.LOSS,
                                "mt_fg_loss": cfg.MT.FG_HINT}
        self.dataloader_s = data_loader["source"]
        self.dataloader_u = data_loader.get("no_label") if cfg.DATASETS.NO_LABEL else None
        self.n_step_unlabel = cfg.MT.N_STEP_UNLABEL
        self.weight_sum_loss = partial(weight_sum_losses, rampup_length=cfg.MT.RAMPUP_STEP,
                                       rampdown_length=cfg.MT.RAMPDOWN_STEP, total_length=self.max_iter,
                                       l=self.lambda_value, balanced=self.balanced_weight, start_mt=self.start_mt)
        self.flat_s = flatten_model(self.student)
        self.flat_t = flatten_model(self.teacher)
        self._unl_iter = None
        # The teacher's no-grad forward and the student's supervised forward are independent until the consistency
        # losses: run the teacher on its own HIP stream (issued from a helper thread -- both forwards contain host syncs),
        # so that its large convolutions fill the GPU while the other stream is in launch-latency-bound target / proposal
        # glue, and vice versa.  MMT_OVERLAP_TEACHER=0 restores the serial order.
        self.overlap_teacher = os.environ.get("MMT_OVERLAP_TEACHER", "1") != "0" and self.device.type == "cuda"
        self.early_sup_backward = True
        # "pair" (one N = 4 forward, two autograd graphs: the default) | "split" (two passes) | "batched" (one pass, one graph):
        # the alternatives are what tests/test_train_step_gpu.py compares the default schedule with
        self.student_passes = "pair"
        self.pair_wgrads = None   # tests: True / False overrides MMT_WGRAD_PAIR
        self.skipped_pairs = 0  # steps whose consistency branch was skipped (no pseudo box on some image)
        # priority -1: HIP maps streams of one priority onto a few hardware queues round-robin; once RCCL has created its own
        # streams (torch.distributed initialised) a default-priority side stream lands on the SAME hardware queue as the
        # main stream and the overlap silently disappears (measured: 63.7 vs 59.9 ms/step).  A different priority class
        # has its own queues.
        self.t_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("MMT_TEACHER_PRIO", "-1"))) if self.overlap_teacher else None
        self._bucketed = None  # BucketedAllReduce, built lazily when enabled (see _bucketed_allreduce)
        # One random stream per model: the teacher's forward runs in a helper thread beside the student's, and with the
        # global generator the interleaving of their draws (fg/bg sampler keys, dropout) would depend on thread timing.
        # With its own generator each model draws the same sequence in the overlapped and in the serial schedule.
        self.gen_s = self.gen_t = None
        if self.device.type == "cuda":
            self.gen_s = torch.Generator(device=self.device)
            self.gen_t = torch.Generator(device=self.device)
            self.seed_rng(torch.initial_seed())
            self.student.set_rng(self.gen_s)
            self.teacher.set_rng(self.gen_t)
        self.teacher_check_period = int(os.environ.get("MMT_TEACHER_CHECK_PERIOD", "100"))

    def seed_rng(self, seed):
        """re-seed the student's and the teacher's random streams (samplers, dropout)"""
        if self.gen_s is not None:
            self.gen_s.manual_seed(int(seed) * 2 + 1)
            self.gen_t.manual_seed(int(seed) * 2 + 2)

    # ---- one iteration (the unit bench.py times)
    def train_step(self, iteration, data_s, target_s, data_u_list=None):
        use_mt = iteration > self.start_mt and self.lambda_value > 0 and data_u_list is not None
        fused._WG_PARKED.clear()   # (jobs a failed step left parked must not meet this step's)
        H.WEIGHTS_GEN[0] += 1      # (what _hip derived from loose weights during the last step is not trusted across steps)
        bucketed = self._bucketed_allreduce()
        if bucketed is not None:
            bucketed.install()  # the stage hooks are registered by the forward passes below
        feats_s = feats_u = None
        early, cut = False, None
        job = None
        if use_mt and self.student_bs == 1:
            xs = data_s.tensors.to(self.device)
            xu = data_u_list[-1].tensors.to(self.device)
            if self.student_passes == "pair" and xs.shape == xu.shape:
                # one set of forward launches for both passes (N = 4), two autograd graphs (modeling/backbone/backbone.py:
                # forward_pair): the schedule below is that of "split"
                from maskrcnn_benchmark.modeling.backbone.backbone import forward_pair
                feats_s, fu = forward_pair(self.student.backbone, xs, xu)
                feats_u = [fu]
                early = True
            elif self.student_passes in ("split", "pair"):
                # Two student backbone passes, labeled crops and unlabeled view.  The supervised branch then runs its WHOLE
                # backward (heads, FPN, backbone) while this thread would otherwise only wait for the teacher -- the teacher is
                # the critical path of the forward and its launch-bound stretches leave the GPU room -- and only the
                # consistency branch is left for after the teacher.  (Batching the two passes makes larger GEMMs but keeps
                # the backbone backward behind the teacher: 50.0 vs 51.8 ms/step.)
                # Measured on the stationary bench (46.4 ms): starting the teacher BEFORE these passes 50.5, the unlabeled
                # view's pass issued after the supervised backward 46.5, batched 48.0, batched + teacher first 49.0 --
                # between 8 and 37 ms both streams hold convolution work and the step is the sum of the kernel times.
                feats_s = self.student.run_backbone(xs, 0)
                feats_u = [self.student.run_backbone(xu, 1)]
                early = True
            elif xs.shape[1:] == xu.shape[1:]:
                # one pass over [labeled crops ; unlabeled student view]; the two forwards consume their slice of the pyramid
                pyr = self.student.backbone(torch.cat([xs, xu], 0))
                n = xs.shape[0]
                parts = [fused.split_batch(l, n) for l in pyr]
                feats_s = tuple(p[0] for p in parts)
                feats_u = [tuple(p[1] for p in parts)]
                if self.early_sup_backward:
                    # graph cut at the pyramid: the heads run backward per branch (the supervised ones before the teacher
                    # is back), backbone + FPN once on the two slices' gradients together
                    roots = feats_s + feats_u[0]
                    feats_s = tuple(t.detach().requires_grad_(True) for t in feats_s)
                    feats_u = [tuple(t.detach().requires_grad_(True) for t in feats_u[0])]
                    cut = (roots, feats_s + feats_u[0])
                    early = True
        if use_mt and self.overlap_teacher and job is None:
            job = self._start_teacher(data_u_list)
        try:
            self.scheduler.step()
            if early:
                self.optimizer.zero_grad()
            loss_dict = self.forward_source(data_s, target_s, feats_s)
            if early:
                defer = _WGRAD_DEFER and use_mt and job is not None
                pairing = (_WGRAD_PAIR if self.pair_wgrads is None else self.pair_wgrads) and use_mt
                from maskrcnn_benchmark.layers import fused as _fused
                gated = defer and _WGRAD_GATE and not pairing
                if gated:
                    defer = False
                    _fused.gate_wgrads(lambda j=job: j.get("backbone_done"))
                if defer:
                    _fused.defer_wgrads(True)
                if pairing:
                    # the supervised pass's weight-gradient jobs are PARKED: those of layers the consistency branch runs through
                    # too (ResNet body, FPN, box head) leave with their partner as one two-segment launch (layers/fused.py)
                    _fused.wgrad_pair_phase("first")
                try:
                    self._backward_roots(loss_dict, iteration)
                    losses_dict = self._weighted_detached(loss_dict, iteration)   # (for the log: behind the backward's launches)
                finally:
                    _fused.wgrad_pair_phase(None)
                    if gated:
                        _fused.gate_wgrads(None)
                        _fused.flush_deferred_wgrads()   # (jobs collected while the teacher's backbone had not been issued yet)
                    if defer:
                        _fused.defer_wgrads(False)
                        _fused.flush_deferred_wgrads()   # one batch, behind the supervised backward, beside the consistency branch
                unl_raw = self.forward_unlabel(data_u_list, feats_u, job)
                unl = unl_raw
                job = None
                if unl:
                    if pairing:
                        _fused.wgrad_pair_phase("second")
                    try:
                        self._backward_roots(unl_raw, iteration)
                        unl = self._weighted_detached(unl_raw, iteration)
                    finally:
                        _fused.wgrad_pair_phase(None)
                if pairing:
                    _fused.finish_wgrad_pairs()   # jobs that found no partner (RPN / mask head; a skipped consistency branch)
                losses_dict.update(unl)
                self._pad_mt_keys(losses_dict)
                if cut is not None:
                    pairs = [(r, l.grad) for r, l in zip(*cut) if l.grad is not None]
                    torch.autograd.backward([r for r, _ in pairs], [g for _, g in pairs])
            else:
                if use_mt:
                    unl = self.forward_unlabel(data_u_list, feats_u, job)
                    job = None
                    loss_dict.update(unl)
                losses_dict = self.weight_sum_loss(loss_dict, iteration)
                self.optimizer.zero_grad()
                sum(v for v in losses_dict.values()).backward()
                if use_mt:
                    self._pad_mt_keys(losses_dict)
        finally:
            if job is not None:  # an exception before the join: never leave the helper thread running on t_stream
                job["thread"].join()
                torch.cuda.current_stream().wait_stream(self.t_stream)
            if bucketed is not None:
                bucketed.finish()
        from maskrcnn_benchmark.layers.fused import join_wgrads
        join_wgrads()   # the weight gradients of this step (side stream) before anything reads the flat gradient
        if bucketed is None:
            allreduce_gradients(self.flat_s)
        sync_touched(self.flat_s)   # every step: a degenerate route on one rank touches another set than its peers'
        # the previous step's EMA reads the student on the teacher's stream: SGD must not overwrite it first.  On a mean-teacher step
        # forward_unlabel's join has ordered the two already; before START_MT (and with no unlabeled batch) nothing else does
        self.sync_teacher()
        self.optimizer.step()
        # round 6: every launch of the step is ordered in front of this point (teacher joined, weight-gradient stream joined): the
        # producing sites' maxima of this step become their plane scales for the next one (_hip.rb_scales_update, one launch)
        H.rb_scales_update()
        if self.lambda_value > 0 and iteration > (self.start_mt - 10):
            self.update_teacher(iteration - (self.start_mt - 10))
            if self.teacher_check_period > 0 and iteration % self.teacher_check_period == 0:
                self.sync_teacher()
                check_teacher_identity(self.flat_t)
        return losses_dict

    def _pad_mt_keys(self, losses):
        """A rank whose teacher found no boxes skips the consistency branch (reference: bare except, MTtrainer.py:258-265)
        and has no mt_* losses this step; the logging reduce (reduce_loss_dict) stacks the values of every key, so all
        ranks must carry the same keys: the skipped ones are reported as zeros (in place)."""
        for k, on in (("mt_fg_loss", self.cfg.MT.FG_HINT), ("mt_classifier", self.cfg.MT.CLS_LOSS)):
            if on and k not in losses:
                losses[k] = next(iter(losses.values())).detach().new_zeros(())
        return losses

    def _bucketed_allreduce(self):
        """the overlapped exchange, when there is somebody to exchange with"""
        # MMT_BUCKETED_ALLREDUCE=0 falls back to the single all-reduce after backward.  (With RCCL at world size 1 the two
        # cost the same, 59.6 vs 60.2 ms/step; what the overlap buys at 2..8 GPUs is the xGMI time of ~3/4 of the 176 MB.)
        if not (dist.is_available() and dist.is_initialized()) or os.environ.get("MMT_BUCKETED_ALLREDUCE", "1") == "0":
            return None
        if self._bucketed is None:
            self._bucketed = BucketedAllReduce(self.flat_s, self.student.backbone.body)
        return self._bucketed

    def train(self):
        self.student.train()
        self.teacher.eval()
        t0 = time.time()
        for iteration, (data_s, target_s, _) in enumerate(self.dataloader_s, self.start_iter):
            data_u = None
            if iteration > self.start_mt and self.lambda_value > 0 and self.dataloader_u is not None:
                data_u = self._next_unlabeled()
            losses_dict = self.train_step(iteration, data_s, target_s, data_u)
            if iteration % 20 == 0 or iteration == self.max_iter:
                red = reduce_loss_dict(losses_dict)
                self.logger.info("iter: %d  %s  lr: %.6f  max mem: %.0f", iteration,
                                 "  ".join("%s: %.4f" % (k, float(v)) for k, v in red.items()),
                                 self.optimizer.param_groups[0]["lr"], torch.cuda.max_memory_allocated() / 2 ** 20)
            if self.ckpt_s is not None and iteration > 0 and iteration % self.checkpoint_period == 0:
                self.save_model(iteration)
        self.logger.info("Total training time: %.1f s", time.time() - t0)

    def _next_unlabeled(self):
        if self._unl_iter is None:
            self._unl_iter = iter(self.dataloader_u)
        try:
            return next(self._unl_iter)[0]
        except StopIteration:
            self._unl_iter = iter(self.dataloader_u)
            return next(self._unl_iter)[0]

    def save_model(self, iteration=0, final=False):
        self.sync_teacher()
        name = "model_final" if final else "model_{:07d}".format(iteration)
        self.ckpt_s.save(name)
        if iteration > self.start_mt and self.ckpt_t is not None:
            self.ckpt_t.save("t_" + name)

    def forward_source(self, image, target, features=None):
        return self.student(image.to(self.device), [t.to(self.device) for t in target], features=features)

    def forward_only(self, data_s, target_s, data_u_list):
        """the three FORWARDS of one iteration and nothing else, without autograd, on the current stream: [A] supervised student
        forward with its losses, [B] the teacher's pseudo labels + K-aug x flip pyramids, [C] the student's consistency losses on
        the unlabeled view -- 2 + 8 + 2 = 12 image-forwards through the backbone at the bench's batch.  What BASELINE.json's
        north_star quotes its 0.5-of-roofline target on (bench.py: forward_leg); results are not used for training."""
        with torch.no_grad():
            out = dict(self.forward_source(data_s, target_s))
            out.update(self.forward_unlabel(data_u_list, None, None))
        return out

    def _loss_coeff(self, k, w):
        return (w if "mt" in k else 1.0) * (self.balanced_weight[k] if k in self.balanced_weight else 1.0)

    def _weighted_detached(self, loss_dict, iteration):
        """the weighted losses of `weight_sum_losses` as detached values (what a step returns for the log), in one multi-tensor launch"""
        w = mt_weight(iteration, self.cfg.MT.RAMPUP_STEP, self.cfg.MT.RAMPDOWN_STEP, self.max_iter, self.lambda_value, self.start_mt)
        keys = [k for k, v in loss_dict.items() if torch.is_tensor(v)]
        out = {k: v for k, v in loss_dict.items() if not torch.is_tensor(v)}
        if keys:
            vals = torch._foreach_mul([loss_dict[k].detach() for k in keys], [float(self._loss_coeff(k, w)) for k in keys])
            out.update(zip(keys, vals))
        return {k: out[k] for k in loss_dict}

    def _backward_roots(self, loss_dict, iteration):
        """backward of sum_k c_k loss_k (c_k = the weights of `weight_sum_losses`, MTtrainer.py:67-109) WITHOUT forming the sum: the
        losses are the roots, the weights their seed gradients.  `sum(weighted.values()).backward()` put ~20 scalar launches (the
        weighting multiplies, the chain of adds, their backward nodes) in front of the first kernel of each backward pass -- on the
        step's critical chain, at a point where the host is not ahead of the device."""
        if not _LOSS_ROOTS:   # (A/B timing: the sum formed by tensor arithmetic, as the reference writes it)
            sum(v for v in self.weight_sum_loss(loss_dict, iteration).values()).backward()
            return
        w = mt_weight(iteration, self.cfg.MT.RAMPUP_STEP, self.cfg.MT.RAMPDOWN_STEP, self.max_iter, self.lambda_value, self.start_mt)
        roots, seeds = [], []
        cache = self.__dict__.setdefault("_seed_cache", {})
        for k, v in loss_dict.items():
            if not (torch.is_tensor(v) and v.requires_grad):
                continue
            c = self._loss_coeff(k, w)
            key = (float(c), v.dtype, v.device, tuple(v.shape))
            t = cache.get(key)
            if t is None:
                if len(cache) > 64:
                    cache.clear()
                t = cache[key] = torch.full(v.shape, float(c), dtype=v.dtype, device=v.device)
            roots.append(v)
            seeds.append(t)
        if roots:
            torch.autograd.backward(roots, seeds)

    def _start_teacher(self, data_u_list):
        """launch teacher.forward_teacher on the side stream from a helper thread; -> job dict (joined in forward_unlabel)"""
        import threading
        teacher_list = [f.to(self.device) for f in data_u_list[:self.teacher_bs]]
        # EMA / weight packing of the previous step, inputs.  (Round 4: letting the side stream start behind the step's FIRST launch
        # instead -- the teacher's N = 8 kernels then run beside the student's N = 4 forward -- was measured SLOWER, medians 38.1 /
        # 38.7 vs 37.6 / 36.1 ms on one box: the student's chain is the critical path and the high-priority side stream starves it.)
        self.t_stream.wait_stream(torch.cuda.current_stream())
        job = {}

        def mark():   # called by the teacher when its K x flip backbone pass has been issued (on the teacher's stream)
            ev = torch.cuda.Event()
            ev.record(self.t_stream)
            job["backbone_done"] = ev

        def run():
            try:
                torch.cuda.set_device(self.device)
                self.teacher.on_backbone_issued = mark
                with torch.cuda.stream(self.t_stream), torch.no_grad():
                    job["result"] = self.teacher.forward_teacher(teacher_list)
            except BaseException as e:  # re-raised (or handled) by the step thread
                job["error"] = e

        job["thread"] = threading.Thread(target=run, name="mmt-teacher", daemon=True)
        job["thread"].start()
        return job

    def forward_unlabel(self, data_u_list, features=None, job=None):
        """MTtrainer.py:247-275 (N_STEP_UNLABEL = 1)"""
        student = [s.to(self.device) for s in data_u_list[-self.student_bs:]]
        emb = None
        if features is not None and len(features) == 1 and self.cfg.MT.FG_HINT and self.cfg.MT.CLS_LOSS and torch.is_grad_enabled():
            # two consumers of every pyramid level (hint adaptor, box pooler): their gradients are summed in one launch
            f_emb, f_box = fused.fork_levels(features[0], 2)
            if job is None:
                emb = self.student.get_emb_feature([f_emb])
            features, f_emb = [f_box], [f_emb]
        else:
            f_emb = features
        if job is not None and features is not None and self.cfg.MT.FG_HINT:
            emb = self.student.get_emb_feature(f_emb)  # independent of the teacher: queued before the wait
        try:
            if job is not None:
                job["thread"].join()
                # everything the teacher produced is read on this stream from here on.  (The side stream IS used again in this
                # step: update_teacher queues the EMA and the teacher's plane re-pack there -- behind the step stream -- and the
                # next step's SGD waits for them through sync_teacher.)  This join also covers an EMA still pending from the last step
                torch.cuda.current_stream().wait_stream(self.t_stream)
                self._teacher_pending = False
                if "error" in job:
                    raise job["error"]
                teacher_results = job["result"]
            else:
                self.sync_teacher()
                teacher_list = [f.to(self.device) for f in data_u_list[:self.teacher_bs]]
                with torch.no_grad():
                    teacher_results = self.teacher.forward_teacher(teacher_list)
        except ValueError as e:  # no pseudo boxes for an image: the reference skips the pair (bare except)
            self.logger.info("teacher produced no boxes (%s), skip this pair", e)
            self.skipped_pairs += 1
            return {}
        return self.student.forward_student(student, teacher_results, features=features, embeddings=emb)

    def update_teacher(self, it):
        """MTtrainer.py:277-281 as one launch over the flat parameter buffers.  With the teacher on its own stream the update and
        the re-packing of the teacher's weight planes run THERE (round 4): nothing on the step stream reads the teacher before the
        next `_start_teacher`, which orders the side stream behind the step stream anyway, so the next step's student forward no
        longer queues behind ~0.3 ms of full-chip HBM passes.  Whoever reads the teacher from outside a step (checkpoints, the
        identity check, tests) goes through `sync_teacher()` or a device-wide synchronize."""
        alpha = min(1 - 1 / (it + 1), self.alpha)
        if self.t_stream is not None and self.overlap_teacher:
            self.t_stream.wait_stream(torch.cuda.current_stream())     # the student's SGD step
            with torch.cuda.stream(self.t_stream):
                H.ema_update(self.flat_t.data, self.flat_s.data, alpha)
                self.flat_t.refresh_planes()
            self._teacher_pending = True
            return
        H.ema_update(self.flat_t.data, self.flat_s.data, alpha)
        self.flat_t.refresh_planes()

    def sync_teacher(self):
        """the step stream waits for a teacher update still queued on the teacher's stream"""
        if getattr(self, "_teacher_pending", False):
            torch.cuda.current_stream().wait_stream(self.t_stream)
            self._teacher_pending = False
