"""Shader clock and board power of THIS GPU (sysfs hwmon of the device torch runs on) while it runs: nothing, the dominant
convolution kernel back to back, a plain HBM copy, the bench step.  The matrix-pipe peak scales with the clock: what the
part sustains under the kernel is the practical ceiling of its MFMA roofline."""
import os, sys, time, glob, threading, torch
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip as H
H.lib()
bus = torch.cuda.get_device_properties(0).pci_bus_id
dev = [d for d in glob.glob("/sys/bus/pci/devices/*") if d.lower().endswith(":%02x:00.0" % bus)]
hw = glob.glob(dev[0] + "/hwmon/hwmon*/") if dev else []
print("device", dev, "hwmon", hw)
def rd(name):
    try: return float(open(hw[0] + name).read())
    except Exception: return float("nan")
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        samples.append((time.perf_counter(), rd("freq1_input") / 1e6, rd("power1_input") / 1e6 if os.path.exists(hw[0] + "power1_input") else rd("power1_average") / 1e6))
        time.sleep(0.005)
def measure(name, fn, seconds=2.0):
    del samples[:]; stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
        if n % 20 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    stop[0] = True; th.join()
    s = [x for x in samples if x[0] > t0 + 0.5]   # after the ramp
    f = sorted(x[1] for x in s); p = sorted(x[2] for x in s)
    print("%-44s %6d calls %8.3f ms each | sclk MHz median %5.0f (p10 %5.0f, p90 %5.0f) | power W median %4.0f" % (
        name, n, (t1 - t0) / max(n, 1) * 1e3, f[len(f) // 2], f[len(f) // 10], f[len(f) * 9 // 10], p[len(p) // 2]))
    return (t1 - t0) / max(n, 1)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
measure("idle", lambda: time.sleep(0.01), 1.5)
x = cl(torch.randn(8, 256, 256, 256, device="cuda")); w = cl(torch.randn(256, 256, 3, 3, device="cuda") * 0.05)
xp = H.split_planes(x)
t = measure("conv3x3_strip_kernel FPN 3x3 256@256^2 N=8", lambda: H.conv_forward(x, w, None, None, 1, 1, x_planes=xp), 3.0)
print("   -> %.1f TFLOP/s algorithmic" % (2.0 * 8 * 256 * 256 * 256 * 256 * 9 / t / 1e12))
prev = H.get_conv_precision(); H.set_conv_precision(0)
t = measure("fp32-input MFMA kernel, same shape", lambda: H.conv_forward(x, w, None, None, 1, 1), 3.0)
print("   -> %.1f TFLOP/s" % (2.0 * 8 * 256 * 256 * 256 * 256 * 9 / t / 1e12))
H.set_conv_precision(prev)
a = torch.empty(1 << 28, device="cuda"); b = torch.empty_like(a)
t = measure("HBM copy 1 GiB -> 1 GiB", lambda: b.copy_(a), 2.0)
print("   -> %.2f TB/s" % (2 * a.numel() * 4 / t / 1e12))
import bench
cfg, trainer, batch = bench.build(torch.device("cuda", 0), 0, base_lr=bench.BENCH_BASE_LR)
it = [1400]
def step():
    il, tg, ul = batch(); trainer.train_step(it[0], il, tg, ul); it[0] += 1
for _ in range(5): step()
measure("bench step (fp32-grade arithmetic)", step, 4.0)
