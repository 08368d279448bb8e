"""The tap-strip kernel's K order (MMT_STRIP_KORDER = 0: kh outermost as in rounds 2-5 | G = 1, 2, 4, 8: groups of G channel slabs
outermost) on the step's large shapes, back to back for 1.5 s each (the part settles at its power cap): ms per call, TFLOP/s, shader
clock.  Results of the orders are compared against order 0 (summation order differs: close, not equal)."""
import os, sys, time, glob, threading, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
H.lib()
bus = torch.cuda.get_device_properties(0).pci_bus_id
dev = [d for d in glob.glob("/sys/bus/pci/devices/*") if d.lower().endswith(":%02x:00.0" % bus)]
hw = glob.glob(dev[0] + "/hwmon/hwmon*/") if dev else []
def rd(name):
    try: return float(open(hw[0] + name).read())
    except Exception: return float("nan")
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        samples.append((time.perf_counter(), rd("freq1_input") / 1e6))
        time.sleep(0.005)
def measure(fn, seconds=1.5):
    del samples[:]; stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
        if n % 20 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    stop[0] = True; th.join()
    f = sorted(x[1] for x in samples if x[0] > t0 + 0.4)
    return (t1 - t0) / n, f[len(f) // 2] if f else float("nan")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
SHAPES = [(8, 256, 256, 256, 256), (2, 256, 256, 256, 256), (4, 256, 256, 256, 256), (8, 256, 128, 128, 256), (8, 128, 128, 128, 128), (2, 256, 128, 128, 256)]
for (N, Cin, Hh, W, Cout) in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = cl(torch.randn(N, Cin, Hh, W, generator=g).relu().cuda()); w = cl((torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda())
    ref = None
    for rep in range(2):
        for order in ("0", "1", "2", "4", "8"):
            os.environ["MMT_STRIP_KORDER"] = order
            y = H.conv_forward(x, w, None, None, 1, 1)
            if ref is None: ref = y.clone()
            err = ((y - ref).abs().max() / ref.abs().max()).item()
            t, clk = measure(lambda: H.conv_forward(x, w, None, None, 1, 1))
            print("N=%d %dx%d %d->%d  order %s: %.4f ms  %.1f TFLOP/s  sclk %4.0f MHz  cycles %.3f M  max rel diff to order 0 %.1e" % (
                N, Hh, W, Cin, Cout, order, t * 1e3, 2.0 * N * Hh * W * Cin * Cout * 9 / t / 1e12, clk, t * clk, err), flush=True)
