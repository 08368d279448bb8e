"""`maskrcnn_benchmark._C` -- the five names the reference's torch extension exports
(csrc/vision.cpp:7-13), bound to libmmtpsm.so instead of the reference's CUDA/CPU kernels.

Contract kept: tensors in / freshly allocated tensors out, NCHW *shapes*, errors as
RuntimeError, launch on the current stream.  Inputs must live on the GPU: like the reference built
without WITH_CUDA raises "Not compiled with GPU support" for CUDA tensors (csrc/nms.h:24), this build
raises for CPU tensors -- there is no CPU implementation in the product.
"""
import torch

from . import _hip
from .utils.miscellaneous import dev_const


def nms(dets, scores, threshold):
    """dets (n,4) xyxy fp32, scores (n,) -> int64 (k,) ascending original indices; CPU-path semantics of the
    reference (cpu/nms_cpu.cpp:37-64: +1 areas, IoU >= thr suppresses)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    _hip._dev(dets, "dets")
    order = torch.sort(scores, descending=True, stable=True)[1]
    n = dets.shape[0]
    seg = dev_const([0, n], torch.int32, dets.device)
    keep, cnt = _hip.nms_batched(dets[order], seg, n, threshold)
    k = int(cnt[0])
    return torch.sort(order[keep[0, :k].long()])[0]


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    lv = torch.zeros((rois.shape[0],), dtype=torch.int32, device=rois.device)
    return _hip.roi_align_forward([input], [spatial_scale], rois, lv, pooled_height, pooled_width, sampling_ratio)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    lv = torch.zeros((rois.shape[0],), dtype=torch.int32, device=rois.device)
    return _hip.roi_align_backward(grad, [(batch_size, channels, height, width)], [spatial_scale], rois, lv,
                                   pooled_height, pooled_width, sampling_ratio)[0]


def roi_pool_forward(*args):
    raise RuntimeError("ROIPool is not part of the R-50-FPN hot path (modeling/poolers.py:66 hard-codes ROIAlign); "
                       "not implemented in the MI355X build")


def roi_pool_backward(*args):
    raise RuntimeError("ROIPool is not part of the R-50-FPN hot path; not implemented in the MI355X build")
