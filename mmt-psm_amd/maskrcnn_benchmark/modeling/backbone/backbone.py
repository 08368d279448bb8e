"""R-50-FPN backbone (reference: modeling/backbone/backbone.py:19-44, resnet.py:61-307, fpn.py:7-74).

Same attribute tree / state-dict keys (backbone.body.stem.*, backbone.body.layerK.N.convM / bnM /
downsample.{0,1}, backbone.fpn.fpn_inner{1-4} / fpn_layer{1-4}).  Execution differs: the whole body is NHWC,
every FrozenBatchNorm + ReLU + residual add lives in a conv epilogue, a bottleneck is 3-4 launches and one
autograd node, the FPN is one autograd node; the frozen stem + layer1 (FREEZE_CONV_BODY_AT=2,
resnet.py:106-115) run without recording anything for backward."""
from collections import OrderedDict

import torch
from torch import nn

from maskrcnn_benchmark import _hip as H
from maskrcnn_benchmark.layers import Conv2d, FrozenBatchNorm2d, fused


import os as _os
_STEM_FUSED = [_os.environ.get("MMT_STEM_FUSED", "1") != "0"]   # (A/B timing, parity tests: [0] = False -> the three-launch stem)


class StemWithFixedBatchNorm(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        out = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.conv1 = Conv2d(3, out, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(out)

    def _s2d_weight(self):
        """the 7x7 / stride-2 filter as a 4x4 / stride-1 filter over the 2x2 space-to-depth image: pad the filter
        to 8x8 with a zero row / column in FRONT (tap kh sits at 2a + b - 1), fold the parity (b_h, b_w) and the
        4-padded RGB channel into 16 input channels.  Cached: the stem is frozen (reference resnet.py:StemWithFixedBatchNorm
        + FREEZE_CONV_BODY_AT >= 1), re-made only if the parameter is written."""
        w = self.conv1.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_s2d_key", None) != key:
            co = w.shape[0]
            w8 = w.new_zeros((co, 8, 8, 4))
            w8[:, 1:, 1:, :3] = w.detach().permute(0, 2, 3, 1)
            ws = w8.view(co, 4, 2, 4, 2, 4).permute(0, 1, 3, 2, 4, 5).reshape(co, 4, 4, 16)
            self._s2d_w = ws.contiguous().permute(0, 3, 1, 2)  # (co, 16, 4, 4) in channels_last memory
            self._s2d_key = key
        return self._s2d_w

    def fused_ok(self, x, count=True):
        """does `x` take the one-launch stem?  (also asked by the detector's run_backbone: only that branch issues nothing but
        C-ABI launches on the image itself, which is what a launch plan can replay -- the two branches below build a re-arranged
        copy of the image with tensor operations a plan never records)"""
        n, c, h, w = x.shape
        return (h % 4 == 0 and w % 4 == 0 and c == 3 and H.F16X2 and H.get_conv_precision() == 3 and not H.bf16_storage()
                and x.dtype == torch.float32 and x.is_contiguous() and self.conv1.out_channels == 64 and _STEM_FUSED[0]
                and H._site_ok(("stem", self.conv1.weight.data_ptr()), x, count=count))

    def forward(self, x):
        n, c, h, w = x.shape
        s, b = self.bn1.folded()
        if self.fused_ok(x):
            # round 5: the whole stem -- convolution, FrozenBN, ReLU, max pool -- as one launch (csrc/conv_stem.hip): the 537 MB of
            # un-pooled output (8 x 1024^2) never exist; bit-identical to the three launches below
            return H.stem_fused(x, self._s2d_weight(), s, b)
        if h % 2 == 0 and w % 2 == 0:
            # 16-channel space-to-depth image -> the DMA-fed split-bf16 kernel instead of the 4-channel fp32 one:
            # out(ho) = sum_kh x(2 ho - 3 + kh) w(kh) = sum_{a,b} z(ho - 2 + a, b) w8(2 a + b),  z(i, b) = x(2 i + b)
            z = x.new_zeros((n, h // 2, w // 2, 2, 2, 4))
            z[..., :3] = x.view(n, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 3, 5, 1)
            y = H.conv_forward(z.view(n, h // 2, w // 2, 16).permute(0, 3, 1, 2), self._s2d_weight(), s, b, 1, 2,
                               relu=True, out_size=(h // 2, w // 2), out_dtype=torch.bfloat16 if H.bf16_storage() else None)
            return _pool_keep_stats(y)
        # odd sizes: pad RGB -> 4 channels (zero weight on the 4th), fp32-input kernel
        x4 = x.new_zeros((n, h, w, 4))
        x4[..., :3] = x.permute(0, 2, 3, 1)
        w4 = self.conv1.weight.new_zeros((self.conv1.out_channels, 7, 7, 4))
        w4[..., :3] = self.conv1.weight.detach().permute(0, 2, 3, 1)
        y = H.conv_forward(x4.permute(0, 3, 1, 2), w4.permute(0, 3, 1, 2), s, b, 2, 3, relu=True,
                           out_dtype=torch.bfloat16 if H.bf16_storage() else None)
        return _pool_keep_stats(y)


def _pool_keep_stats(y):
    """3x3 / stride-2 / pad-1 max pooling of a ReLU output.  Every element of y lies in some window and y >= 0, so
    max |pool(y)| == max |y| exactly: the statistics slot the stem convolution recorded for y serves the pooled tensor too
    (its sampled mean is y's, i.e. lower than the pooled tensor's: the crest-factor test errs on the careful side) and
    layer1's first convolutions need no reduction pass over the 134 MB tensor."""
    p = H.maxpool3x3s2(y)
    am = getattr(y, "_mmt_amax", None)
    if am is not None and am[1] == y._version:
        p._mmt_amax = (am[0], p._version)
    return p


class BottleneckWithFixedBatchNorm(nn.Module):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True, stride=1):
        super().__init__()
        if num_groups != 1 or not stride_in_1x1:
            raise NotImplementedError("R-50 hot path: NUM_GROUPS=1, STRIDE_IN_1X1=True")
        self.downsample = None
        if in_channels != out_channels:
            self.downsample = nn.Sequential(Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False),
                                            FrozenBatchNorm2d(out_channels))
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride, bias=False)
        self.bn1 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out_channels)
        self.stride = stride

    def _args(self):
        # (once per block and pass, on the launch-issuing thread: sub-modules and parameters straight from the module dicts)
        m = self._modules
        s1, b1 = m["bn1"].folded()
        s2, b2 = m["bn2"].folded()
        s3, b3 = m["bn3"].folded()
        wd = sd = bd = None
        ds = m["downsample"] if "downsample" in m else self.downsample
        if ds is not None:
            wd = ds[0]._parameters["weight"]
            sd, bd = ds[1].folded()
        return (m["conv1"]._parameters["weight"], m["conv2"]._parameters["weight"], m["conv3"]._parameters["weight"], wd,
                (s1, b1, s2, b2, s3, b3, sd, bd), self.stride)

    def forward(self, x, pre=None):
        return fused.BottleneckFn.apply(x, *self._args(), pre)

    def forward_raw(self, x):
        """no autograd: (o1, o2, out) of this block for `x` (forward_pair)"""
        w1, w2, w3, wd, bn, stride = self._args()
        return fused.bottleneck_forward(H.nhwc(x), w1, w2, w3, wd, bn, stride)   # (the caller runs this under no_grad)


class ResNet(nn.Module):
    BLOCKS = {"R-50-FPN": (3, 4, 6, 3), "R-101-FPN": (3, 4, 23, 3)}

    def __init__(self, cfg):
        super().__init__()
        self.stem = StemWithFixedBatchNorm(cfg)
        in_ch = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        width = cfg.MODEL.RESNETS.NUM_GROUPS * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
        out2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
        self.stages = []
        for i, n in enumerate(self.BLOCKS[cfg.MODEL.BACKBONE.CONV_BODY], 1):
            f = 2 ** (i - 1)
            blocks, stride = [], (2 if i > 1 else 1)
            for _ in range(n):
                blocks.append(BottleneckWithFixedBatchNorm(in_ch, width * f, out2 * f, 1,
                                                           cfg.MODEL.RESNETS.STRIDE_IN_1X1, stride))
                stride, in_ch = 1, out2 * f
            self.add_module("layer%d" % i, nn.Sequential(*blocks))
            self.stages.append("layer%d" % i)
        # set by the training engine: callable(stage_name, "registered" | "fired").  A tensor hook on every trainable
        # stage output reports when the gradient of that output is final, i.e. when the backward of everything that
        # consumes it (the next stage, the FPN, the heads) has finished -- the cue for the bucketed gradient all-reduce
        self.grad_ready = None
        self.freeze_at = cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT
        for si in range(self.freeze_at):
            m = self.stem if si == 0 else getattr(self, "layer%d" % si)
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        outs = []
        with torch.no_grad():
            x = self.stem(x)
        for i, name in enumerate(self.stages, 1):
            if i < self.freeze_at:
                with torch.no_grad():
                    x = getattr(self, name)(x)
            else:
                x = getattr(self, name)(x)
                _stage_hook(self, name, x)
            if i >= self.freeze_at and i < len(self.stages):
                o, x = fused.fork(x, 2, ("gC", i) if fused.RB_WIDE else None)   # the FPN lateral and the next stage: their gradients are summed in one launch
                outs.append(o)
            else:
                outs.append(x)
        return outs


def _stage_hook(body, name, x):
    cb = body.grad_ready
    if cb is not None and x.requires_grad:
        cb(name, "registered")
        x.register_hook(lambda g, name=name, cb=cb: cb(name, "fired"))


def forward_pair(backbone, xa, xb):
    """backbone(xa), backbone(xb) for two equally shaped batches that need SEPARATE autograd graphs (the labeled and the
    unlabeled student pass of a mean-teacher step: the supervised backward runs before the consistency branch exists) with
    ONE set of forward launches on the concatenated batch: N = 4 instead of 2 x N = 2 fills the chip better on the few-tile
    layers (fewer split-K ranges, fewer finish launches).  The forward runs without autograd; each half then gets its own
    chain of the same nodes (`pre=`: the node only records the tensors its backward reads, all of them dense views of the
    N = 4 results).  -> (pyramid of xa, pyramid of xb), each exactly what backbone(x) returns."""
    body, fpn = backbone.body, backbone.fpn
    n = xa.shape[0]
    halves = ((0, n), (n, 2 * n))
    with torch.no_grad():
        fused._PAIR_FWD[0] = True
        x = body.stem(torch.cat([xa, xb], 0))
        raw = {}     # block -> (o1, o2, out) on the concatenated batch
        frozen = []
        for i, name in enumerate(body.stages, 1):
            for bi, blk in enumerate(getattr(body, name)):
                if i < body.freeze_at:
                    x = blk(x)
                else:
                    raw[(name, bi)] = blk.forward_raw(x)
                    x = raw[(name, bi)][2]
            if i < body.freeze_at:
                frozen.append(x)
        cs_cat = frozen + [raw[(name, len(getattr(body, name)) - 1)][2] for i, name in enumerate(body.stages, 1) if i >= body.freeze_at]
        wi = [getattr(fpn, nm).weight for nm in fpn.inner_blocks]
        bi_ = [getattr(fpn, nm).bias for nm in fpn.inner_blocks]
        wl = [getattr(fpn, nm).weight for nm in fpn.layer_blocks]
        bl = [getattr(fpn, nm).bias for nm in fpn.layer_blocks]
        # (P_k's planes for the FIRST half only: the RPN head's 3x3 runs on the labeled pass; the unlabeled half's levels feed the box
        # pooler and the hint adaptors, which read fp32 -- _hip.RbLead)
        op = getattr(fpn, "out_planes", True)
        inner_cat, outs_cat = fused.fpn_forward([H.nhwc(c) for c in cs_cat], wi, bi_, wl, bl, n if op is True else op)
        fused._PAIR_FWD[0] = False
    res = []
    for lo, hi in halves:
        outs = []
        x = None
        for i, name in enumerate(body.stages, 1):
            if i < body.freeze_at:
                x = fused.batch_slice(frozen[i - 1], lo, hi)
            else:
                for bi, blk in enumerate(getattr(body, name)):
                    x = blk(x, pre=tuple(fused.batch_slice(t, lo, hi) for t in raw[(name, bi)]))
                _stage_hook(body, name, x)
                if i < len(body.stages):
                    o, x = fused.fork(x, 2, ("gC", i) if fused.RB_WIDE else None)   # FPN lateral / next stage
                    outs.append(o)
                    continue
            outs.append(x)
        args = list(outs)
        for nm in fpn.inner_blocks:
            m = getattr(fpn, nm)
            args += [m.weight, m.bias]
        for nm in fpn.layer_blocks:
            m = getattr(fpn, nm)
            args += [m.weight, m.bias]
        pre = ([fused.batch_slice(t, lo, hi) for t in inner_cat], [fused.batch_slice(t, lo, hi) for t in outs_cat])
        pyr = list(fused.FPNFn.apply(*args, getattr(fpn, "out_planes", True), pre))
        for p_ in pyr:
            _stage_hook(body, "heads", p_)   # fires when the heads' backward has delivered this level's gradient
        for p_, o_ in zip(pyr, pre[1]):  # the node's outputs are new tensor objects: the planes / statistics of the slices go along
            pl = H.planes_of(o_)
            if pl is not None:
                p_._mmt_planes = (pl, p_._version)
            am = getattr(o_, "_mmt_amax", None)
            if am is not None and am[1] == o_._version:
                p_._mmt_amax = (am[0], p_._version)
        if fpn.top_blocks is not None:
            pyr.extend(fpn.top_blocks(pyr[-1]))
        res.append(tuple(pyr))
    return res[0], res[1]


class FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels, top_blocks=None):
        super().__init__()
        self.inner_blocks, self.layer_blocks = [], []
        for idx, c in enumerate(in_channels_list, 1):
            inner, layer = Conv2d(c, out_channels, 1), Conv2d(out_channels, out_channels, 3, 1, 1)
            for m in (inner, layer):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
            self.add_module("fpn_inner%d" % idx, inner)
            self.add_module("fpn_layer%d" % idx, layer)
            self.inner_blocks.append("fpn_inner%d" % idx)
            self.layer_blocks.append("fpn_layer%d" % idx)
        self.top_blocks = top_blocks

    def forward(self, x):
        args = list(x)
        for n in self.inner_blocks:
            m = getattr(self, n)
            args += [m.weight, m.bias]
        for n in self.layer_blocks:
            m = getattr(self, n)
            args += [m.weight, m.bias]
        # out_planes: the pyramid levels go to the RPN head's 3x3 convolution as they are (False for the teacher, whose
        # RPN head sees one view's slice of the batched pyramid)
        res = list(fused.FPNFn.apply(*args, getattr(self, "out_planes", True)))
        body = self.__dict__.get("_body_ref")   # (a plain reference, not a registered sub-module: build_resnet_fpn_backbone)
        if body is not None:
            for t in res:
                _stage_hook(body, "heads", t)   # fires when the heads' backward has delivered this level's gradient
        if self.top_blocks is not None:
            res.extend(self.top_blocks(res[-1]))
        return tuple(res)


class LastLevelMaxPool(nn.Module):
    def forward(self, x):
        y = x[:, :, ::2, ::2]  # max_pool2d(kernel 1, stride 2) == subsampling (fpn.py:72-74)
        am = getattr(x, "_mmt_amax", None)
        if am is not None and am[1] == x._version and x.is_cuda:
            # dense here (the copy its consumers would make anyway), with P5's statistics slot: the maximum of a subset is bounded
            # by it -- what the fp16 split's scale needs (no reduction pass over P6)
            y = y.contiguous(memory_format=torch.channels_last)
            H.record_torch(lambda src=x, dst=y: dst.copy_(src[:, :, ::2, ::2]))   # (a launch plan replays C-ABI calls: this copy too)
            y._mmt_amax = (am[0], y._version)
        return [y]


def build_resnet_fpn_backbone(cfg):
    body = ResNet(cfg)
    c2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    fpn = FPN([c2, c2 * 2, c2 * 4, c2 * 8], cfg.MODEL.BACKBONE.OUT_CHANNELS, LastLevelMaxPool())
    fpn.__dict__["_body_ref"] = body   # where the training engine hangs its `grad_ready` callback (not a sub-module of the FPN)
    return nn.Sequential(OrderedDict([("body", body), ("fpn", fpn)]))


def build_backbone(cfg):
    if cfg.MODEL.BACKBONE.CONV_BODY not in ResNet.BLOCKS:
        raise KeyError("cfg.MODEL.BACKBONE.CONV_BODY: {} is not on the MI355X hot path".format(
            cfg.MODEL.BACKBONE.CONV_BODY))
    return build_resnet_fpn_backbone(cfg)
