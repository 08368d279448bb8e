"""The operator API of the reference (`maskrcnn_benchmark.layers`, layers/__init__.py:4-15): same names, same
call signatures, HIP underneath."""
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from .. import _C
from .. import _hip
from . import fused

nms = _C.nms


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (layers/batch_norm.py:6-24).  NO epsilon:
    scale = weight * rsqrt(running_var).  On the hot path the (scale, shift) pair is folded into the epilogue
    of the preceding convolution; `forward` exists for stand-alone use."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._folded = None

    def folded(self):
        # (asked once per convolution launch: the buffers are read from the module's dict, not through nn.Module.__getattr__)
        bf = self._buffers
        w, b, m, v = bf["weight"], bf["bias"], bf["running_mean"], bf["running_var"]
        key = (w._version, b._version, m._version, v._version, w.data_ptr(), w.device)
        f = self._folded
        if f is None or f[0] != key:
            scale = w * v.rsqrt()
            shift = b - m * scale
            f = self._folded = (key, scale.contiguous(), shift.contiguous())
        return f[1], f[2]

    def forward(self, x):
        scale, shift = self.folded()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class Conv2d(nn.Module):
    """Drop-in for layers.Conv2d / nn.Conv2d on the path (layers/misc.py:30-43): parameter `weight`
    (Cout,Cin,KH,KW) -- kept in channels_last memory = the kernel's [Cout][KH][KW][Cin] -- and optional `bias`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        k = _pair(kernel_size)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, _pair(stride), _pair(padding)
        w = torch.empty(out_channels, in_channels, k[0], k[1])
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, relu=False, input_relu=False, out_rb=False, din_rb=False):
        if x.numel() == 0:  # empty-batch path of the reference (_NewEmptyTensorOp)
            h = (x.shape[2] + 2 * self.padding[0] - self.kernel_size[0]) // self.stride[0] + 1
            w = (x.shape[3] + 2 * self.padding[1] - self.kernel_size[1]) // self.stride[1] + 1
            return x.new_empty((x.shape[0], self.out_channels, h, w))
        pr = self._parameters   # (not through nn.Module.__getattr__: once per launch on the issuing thread)
        return fused.conv(x, pr["weight"], pr.get("bias"), self.stride[0], self.padding[0], relu, input_relu, out_rb, din_rb)

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, padding={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding)


class ConvTranspose2d(nn.Module):
    """2x2 stride-2 transposed convolution of the mask predictor (layers/misc.py:46-64)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        if _pair(kernel_size) != (2, 2) or _pair(stride) != (2, 2) or _pair(padding) != (0, 0):
            raise NotImplementedError("the hot path only has the 2x2 stride-2 deconvolution")
        self.in_channels, self.out_channels = in_channels, out_channels
        w = torch.empty(in_channels, out_channels, 2, 2)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, relu=False, input_relu=False):
        if x.numel() == 0:
            return x.new_empty((x.shape[0], self.out_channels, 2 * x.shape[2], 2 * x.shape[3]))
        pr = self._parameters
        return fused.DeconvFn.apply(x, pr["weight"], pr.get("bias"), relu, input_relu)


class Linear(nn.Module):
    """nn.Linear of the box head (box_head/roi_box_feature_extractors.py:97-98) on the MFMA GEMM"""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        w = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(w, a=1)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(out_features))

    def forward(self, x, relu=False, input_relu=False, in_mask_scale=1.0, mul=None):
        pr = self._parameters
        return fused.linear(x, pr["weight"], pr.get("bias"), relu, input_relu, in_mask_scale, mul)


class _ROIAlign(torch.autograd.Function):
    """Same autograd contract as layers/roi_align.py:11-44: saves the rois only, returns
    (grad_input, None, None, None, None)."""

    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        return _C.roi_align_forward(input, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1], sampling_ratio)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        g = _C.roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                  bs, ch, h, w, ctx.sampling_ratio)
        return g, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "{}(output_size={}, spatial_scale={}, sampling_ratio={})".format(
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)


def roi_pool(*args, **kwargs):
    return _C.roi_pool_forward(*args)


class ROIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale

    def forward(self, input, rois):
        return roi_pool(input, rois, self.output_size, self.spatial_scale)


def smooth_l1_loss(input, target, beta=1. / 9, size_average=True, reduction="sum"):
    """layers/smooth_l1_loss.py:6-20 (a few hundred elements per call: plain device tensor arithmetic)"""
    n = torch.abs(input - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if size_average:
        return loss.mean()
    return loss.sum() if reduction == "sum" else loss.sum(1)


def interpolate(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
    """layers/misc.py:67-102.  On the hot path the nearest x2 upsample of the FPN is fused into the lateral
    conv epilogue (fused.FPNFn); this stand-alone form is kept for API parity."""
    return torch.nn.functional.interpolate(input, size, scale_factor, mode, align_corners)


__all__ = ["nms", "roi_align", "ROIAlign", "roi_pool", "ROIPool", "smooth_l1_loss", "Conv2d", "ConvTranspose2d",
           "interpolate", "FrozenBatchNorm2d", "Linear"]
