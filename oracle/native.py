"""ORACLE -- test infrastructure only (see oracle/__init__.py).

ctypes binding of oracle/native/mmt_oracle.c plus torch-tensor convenience wrappers with
the REFERENCE's `_C` signatures (csrc/vision.cpp:7-13): nms, roi_align_forward,
roi_align_backward -- all CPU, NCHW.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "native", "mmt_oracle.c")
_SO = os.path.join(_HERE, "native", "libmmt_oracle.so")
_lib = None


def build(force=False):
    """gcc -O2, no FMA contraction (the reference extension is built without -march flags)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-std=c99", "-shared", "-fPIC",
                               _SRC, "-o", _SO, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        P = ctypes.c_void_p
        L.orc_nms.restype = ctypes.c_int64
        L.orc_nms.argtypes = [P, P, ctypes.c_int64, ctypes.c_float, P]
        for suf, ft in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            for d in ("forward", "backward"):
                f = getattr(L, "orc_roi_align_%s_%s" % (d, suf))
                f.restype = None
                f.argtypes = [P, P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ft,
                              ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
        L.orc_poly_mask.restype = None
        L.orc_poly_mask.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
        _lib = L
    return _lib


def nms(dets, scores, thr):
    """-> int64[k] ascending original indices (csrc/nms.h:10-28 CPU branch)."""
    assert dets.device.type == "cpu"
    n = dets.shape[0]
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    d = dets.detach().float().contiguous()
    s = scores.detach().float().contiguous()
    keep = torch.empty(n, dtype=torch.int64)
    k = lib().orc_nms(d.data_ptr(), s.data_ptr(), n, float(thr), keep.data_ptr())
    return keep[:k].clone()


def _suf(t):
    assert t.dtype in (torch.float32, torch.float64)
    return "f32" if t.dtype == torch.float32 else "f64"


def roi_align_forward(inp, rois, scale, ph, pw, sr):
    assert inp.device.type == "cpu"
    x = inp.detach().contiguous()
    r = rois.detach().to(x.dtype).contiguous()
    K = r.shape[0]
    N, C, H, W = x.shape
    out = torch.empty(K, C, ph, pw, dtype=x.dtype)
    if out.numel():
        getattr(lib(), "orc_roi_align_forward_" + _suf(x))(
            x.data_ptr(), r.data_ptr(), K, C, H, W, float(scale), ph, pw, sr, out.data_ptr())
    return out


def roi_align_backward(grad, rois, scale, ph, pw, N, C, H, W, sr):
    g = grad.detach().contiguous()
    r = rois.detach().to(g.dtype).contiguous()
    gin = torch.zeros(N, C, H, W, dtype=g.dtype)
    if g.numel():
        getattr(lib(), "orc_roi_align_backward_" + _suf(g))(
            g.data_ptr(), r.data_ptr(), r.shape[0], C, H, W, float(scale), ph, pw, sr, gin.data_ptr())
    return gin


class _RoiAlignFn(torch.autograd.Function):
    """Same autograd contract as layers/roi_align.py:11-44."""

    @staticmethod
    def forward(ctx, inp, rois, out_size, scale, sr):
        ctx.save_for_backward(rois)
        ctx.cfgv = (out_size, scale, sr, inp.shape)
        return roi_align_forward(inp, rois, scale, out_size[0], out_size[1], sr)

    @staticmethod
    def backward(ctx, g):
        rois, = ctx.saved_tensors
        out_size, scale, sr, shp = ctx.cfgv
        return roi_align_backward(g, rois, scale, out_size[0], out_size[1], *shp, sr), None, None, None, None


def roi_align(inp, rois, out_size, scale, sr):
    return _RoiAlignFn.apply(inp, rois, tuple(out_size), scale, sr)


def poly_mask(polys, h, w):
    """polys: list of 1-D float arrays [x0,y0,x1,y1,...] of ONE instance -> uint8 (h,w).

    == mask_utils.decode(mask_utils.merge(mask_utils.frPyObjects(polys, h, w)))
    (structures/segmentation_mask.py:122-133)."""
    xy = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1) for p in polys])
    lens = np.asarray([len(p) // 2 for p in polys], dtype=np.int32)
    out = np.zeros((h, w), dtype=np.uint8)
    lib().orc_poly_mask(xy.ctypes.data, lens.ctypes.data, len(polys), h, w, out.ctypes.data)
    return out
