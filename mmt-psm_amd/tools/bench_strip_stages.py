"""conv3x3_strip_kernel on the two-term fp16 split: two-stage ring (MMT_STRIP_STAGES=2, rounds 2-3) against the three-stage ring
(round 4) on the shapes of a step -- kernel time back to back (the split pass and the weight planes prepared once), and
bit-equality of the two results (same products in the same order)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from maskrcnn_benchmark import _hip as H
L = H.lib()
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(f, it=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def run(x, w, scale, shift, pre):
    N, Cin, Hh, W = x.shape; Cout = w.shape[0]
    (xp, sx), (wp, sw) = pre
    a = H._conv_args(x, w, 1, 1, Hh, W)
    y = H.empty_nhwc(N, Cout, Hh, W, x.device)
    a.y, a.scale, a.shift, a.relu = y.data_ptr(), H._p(scale), H._p(shift), 1
    a.x_planes, a.x_plane_stride, a.w_planes, a.w_plane_stride = xp.data_ptr(), xp.stride(0), wp.data_ptr(), wp.stride(0)
    H._check(L.mmt_conv3x3_strip_f16x2(ctypes.byref(a), sx.data_ptr(), sw.data_ptr(), H._stream()), "strip f16x2")
    return y


def main():
  g = torch.Generator().manual_seed(0)
  cases = [("fpn 256@256^2 N8", 8, 256, 256, 256, 256), ("fpn 256@256^2 N4", 4, 256, 256, 256, 256), ("fpn 256@256^2 N2", 2, 256, 256, 256, 256),
           ("rpn 256@128^2 N8", 8, 256, 128, 128, 256), ("rpn 256@128^2 N2", 2, 256, 128, 128, 256),
           ("l2 128@128^2 N8", 8, 128, 128, 128, 128), ("l2 128@128^2 N2 (split-K)", 2, 128, 128, 128, 128),
           ("l3 256@64^2 N8", 8, 256, 64, 64, 256), ("l3 256@64^2 N4 (split-K)", 4, 256, 64, 64, 256), ("l3 256@64^2 N2 (split-K)", 2, 256, 64, 64, 256)]
  for name, N, Cin, Hh, W, Cout in cases:
      x = cl(torch.randn((N, Cin, Hh, W), generator=g).relu_().cuda())
      w = cl((torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda())
      sc = (torch.rand(Cout, generator=g) + 0.5).cuda(); sh = (torch.randn(Cout, generator=g) * 0.1).cuda()
      pre = (H.f16_split(x), H.f16_weight_planes(w))
      fl = 2.0 * N * Hh * W * Cout * Cin * 9
      out = {}
      for var in ("0", "1", "11", "14", "15"):   # base | every copy behind tap 2 | no copies in the loop | no waits / barriers | neither
          os.environ["MMT_STRIP_VARIANT"] = var
          out[var] = (run(x, w, sc, sh, pre), timeit(lambda: run(x, w, sc, sh, pre)))
      os.environ["MMT_STRIP_VARIANT"] = "0"
      ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), None, 1, 1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
      err = ((out["1"][0][:1].double() - ref.relu()).abs().max() / ref.abs().max()).item()
      print("%-28s base %7.3f ms %6.1f TF | early copies %7.3f ms %6.1f TF (%+5.1f %%) bit-equal %s err %.2e | no copies %7.3f | no waits %7.3f | neither %7.3f" % (
          name, out["0"][1], fl / out["0"][1] / 1e9, out["1"][1], fl / out["1"][1] / 1e9, 100 * (out["0"][1] / out["1"][1] - 1),
          torch.equal(out["0"][0], out["1"][0]), err, out["11"][1], out["14"][1], out["15"][1]), flush=True)


if __name__ == "__main__":
    main()
