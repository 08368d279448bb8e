"""Optimiser / schedule (reference: solver/build.py:5-34, solver/lr_scheduler.py:10-53).

Same arithmetic as torch.optim.SGD with one parameter group per tensor (bias: lr x BIAS_LR_FACTOR, no weight
decay), but executed as two launches of `mmt_sgd_momentum` over a model's FLAT parameter storage
(engine/flat.py): [trainable weights | trainable biases | frozen]."""
from bisect import bisect_right

from maskrcnn_benchmark import _hip as H


class FlatSGD(object):
    def __init__(self, flat, base_lr, momentum, weight_decay, bias_lr_factor, weight_decay_bias):
        self.flat = flat
        self.base_lr, self.momentum = base_lr, momentum
        self.weight_decay, self.bias_lr_factor, self.weight_decay_bias = weight_decay, bias_lr_factor, weight_decay_bias
        self.lr_factor = 1.0
        self.steps = 0
        # mirrors torch's param_groups enough for the trainer's logging line (MTtrainer.py:217)
        self.param_groups = [{"lr": base_lr}]

    def zero_grad(self):
        self.flat.grad.zero_()
        self.flat.touched.clear()

    def state_dict(self):
        """what Checkpointer.save stores under "optimizer": the momentum of every trainable parameter by name (the
        flat layout is an implementation detail of this build and not part of the file)"""
        f = self.flat
        named = dict(f._named)
        buf = {}
        for n, (o, k) in f.index.items():
            if o < f.n_trainable:
                buf[n] = f._view_like(f.momentum[o:o + k], named[n]).detach().cpu().contiguous()  # the parameter's shape
        return {"momentum_buffers": buf, "steps": self.steps, "lr_factor": self.lr_factor,
                "hyper": {"base_lr": self.base_lr, "momentum": self.momentum, "weight_decay": self.weight_decay,
                          "bias_lr_factor": self.bias_lr_factor, "weight_decay_bias": self.weight_decay_bias}}

    def load_state_dict(self, sd):
        f = self.flat
        if "momentum_buffers" not in sd:
            raise ValueError("FlatSGD.load_state_dict: not a FlatSGD state (torch.optim.SGD's state / param_groups format of a "
                             "reference checkpoint is not convertible by name; only the `model` entry is interchangeable)")
        for n, v in sd["momentum_buffers"].items():
            if n in f.index and f.index[n][0] < f.n_trainable:
                o, k = f.index[n]
                f._view_like(f.momentum[o:o + k], dict(f._named)[n]).copy_(v.to(f.momentum.device))
        self.steps = int(sd.get("steps", 0))
        self.lr_factor = float(sd.get("lr_factor", self.lr_factor))

    def step(self):
        """torch.optim.SGD.step over the reference's per-tensor groups (solver/build.py:5-23): a parameter that received no
        gradient since zero_grad (`p.grad is None` there: the hint adaptors before START_MT or when the teacher found nothing,
        heads of a branch that did not run) is left alone -- no weight decay, no momentum decay -- and names containing
        'box_heads.box.D' are never optimised (:11).  The parameters that did receive one form a few contiguous runs of the
        flat buffer: one launch per run.  Momentum buffers start at zero, so torch's lazy `buf = d_p` first step is the
        same arithmetic as `buf = momentum * 0 + d_p`."""
        f = self.flat
        lr = self.base_lr * self.lr_factor
        nw, nb = f.n_weights, f.n_biases
        names = [n for n in f.touched if "box_heads.box.D" not in n]
        for a, b in f.active_ranges(names, 0, nw):
            H.sgd_momentum(f.data[a:b], f.grad[a:b], f.momentum[a:b], lr, self.weight_decay, self.momentum, False)
        for a, b in f.active_ranges(names, nw, nw + nb):
            H.sgd_momentum(f.data[a:b], f.grad[a:b], f.momentum[a:b], lr * self.bias_lr_factor,
                           self.weight_decay_bias, self.momentum, False)
        f.refresh_planes()
        self.steps += 1
        self.param_groups[0]["lr"] = lr


class WarmupMultiStepLR(object):
    """lr_scheduler.py:10-53 (get_lr formula), driving FlatSGD.lr_factor"""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.last_epoch = last_epoch
        self.step()

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "milestones": list(self.milestones), "gamma": self.gamma,
                "warmup_factor": self.warmup_factor, "warmup_iters": self.warmup_iters, "warmup_method": self.warmup_method}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self.optimizer.lr_factor = self.factor()

    def factor(self):
        w = 1
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                w = self.warmup_factor
            else:
                a = self.last_epoch / self.warmup_iters
                w = self.warmup_factor * (1 - a) + a
        return w * self.gamma ** bisect_right(self.milestones, self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr_factor = self.factor()


def make_optimizer(cfg, model):
    from maskrcnn_benchmark.engine.flat import flatten_model
    flat = flatten_model(model)
    s = cfg.SOLVER
    return FlatSGD(flat, s.BASE_LR, s.MOMENTUM, s.WEIGHT_DECAY, s.BIAS_LR_FACTOR, s.WEIGHT_DECAY_BIAS)


def make_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    return WarmupMultiStepLR(optimizer, s.STEPS, s.GAMMA, warmup_factor=s.WARMUP_FACTOR, warmup_iters=s.WARMUP_ITERS,
                             warmup_method=s.WARMUP_METHOD)
