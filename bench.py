#!/usr/bin/env python
"""Benchmark of the MMT-PSM mean-teacher step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of engine/MTtrainer.py on ONE per-GPU batch of synthetic crops already resident
in HBM: [A] supervised student forward on 2 labeled 1000x1000 crops, [B] teacher.forward_teacher on 2 unlabeled
crops x AUG_K=2 views x flip (+ coarse inference on view 0's pyramid), [C] student.forward_student on the AUG_S=1 view (MGD + PSM),
[D] weighted loss, backward, (RCCL all-reduce of the flat student gradient when N > 1), SGD, [E] EMA teacher.
img = one source crop consumed per step (2 labeled + 2 unlabeled = 4 per GPU per step); weak scaling.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the large-tile convolution kernel with the most
event-bracketed time in a separate leg: conv3x3_strip_kernel or conv_fwd_glds_kernel<128,128>, on the matrix pipe through
the two-term fp16 split -- 3 MFMA products per multiply, peak 2500 / 3 TFLOP/s; the 6-product line of the 3-term bf16
split and the fp32-input MFMA peak are quoted beside it) and `cpu_baseline` (the CPU oracle timed on this node's host
cores on a bounded sample; baseline only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table: bf16 dense (v_mfma_f32_32x32x16_bf16)
PRODUCTS = {0: 1, 1: 1, 2: 3, 3: 6}  # MFMA products per algorithmic multiply-add in each arithmetic mode
CROP = 1000
N_LAB, N_UNLAB, N_INST = 2, 2, 12


# The bench runs on noise images with random-init weights.  At the recipe's BASE_LR the detector un-learns its (random)
# foreground scores within ~15 iterations, the teacher then finds NO box on some unlabeled image for stretches of tens of
# iterations, and the trainer -- like the reference's bare `except` (MTtrainer.py:247-275) -- skips the whole consistency
# branch of such a step: 36 instead of 47 ms, i.e. work missing from the timed region (seen as a bimodal per-step series,
# mmt-psm_amd/tools/step_series.py).  A real run past START_MT has detections on every image.  The bench therefore
# freezes the drift: the same optimizer / EMA kernels run on the same bytes every step, with a learning rate that keeps
# the weights where the data-dependent sizes (proposals, detections, pseudo instances) are those of step 0, and it
# counts the steps whose consistency branch was skipped (must be 0) in the JSON line.
BENCH_BASE_LR = 1e-7


def build(device, rank, irnet=False, crop=None, n_inst=None, base_lr=None, n_lab=None, n_unlab=None, supervised=False):
    import synthetic
    from maskrcnn_benchmark.config import make_default_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.solver import make_optimizer, make_lr_scheduler
    from maskrcnn_benchmark.engine.MTtrainer import MTtrainer, init_teacher_weight
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    from maskrcnn_benchmark.structures.image_list import to_image_list

    cfg = make_default_cfg()
    if irnet:  # BASELINE configs[4] (in fp32): learned duplicate removal + mask relation on
        # RELATION_NMS.LOSS 0.01 (shipped: 1.0): the synthetic fc7 features have |x|^2 ~ 1e3, so an MSE head on top of
        # them is unstable at BASE_LR 0.005 with weight 1 (lr * |x|^2 > 2) and the run turns NaN within 4 steps;
        # the loss weight only scales the gradient -- the launches per step are the same
        parts = os.environ.get("MMT_IRNET_PARTS", "nms,mask")   # tools: one of the two relation modules alone
        cfg.merge_from_list(["MODEL.RELATION_NMS.USE_RELATION_NMS", "nms" in parts, "MODEL.RELATION_MASK.USE_RELATION", "mask" in parts,
                             "MODEL.RELATION_NMS.LOSS", 0.01])
    if base_lr is not None:
        cfg.merge_from_list(["SOLVER.BASE_LR", base_lr])
    if supervised:   # BASELINE configs[1]: MT.LAMBDA 0 -- no teacher pass, no consistency branch, no EMA
        cfg.merge_from_list(["MT.LAMBDA", 0.0])
    torch.manual_seed(0)
    student = build_detection_model(cfg, is_student=True)
    teacher = build_detection_model(cfg, is_teacher=True)
    shapes = {k: tuple(v.shape) for k, v in student.state_dict().items()}
    sd = synthetic.make_weights(shapes, seed=0)  # identical on every rank (DP replicas start equal)
    if irnet and student.relation_nms is not None:
        # the relation-NMS IoU regressor keeps its own (seeded) training initialisation: with the O(1) synthetic
        # weights its MSE loss starts at ~20 and diverges to NaN within three SGD steps at BASE_LR.  Its output bias
        # is set to 0.5 (a regressor that predicts IoU 0.5 +- small for every ranked box) so that the teacher's
        # learned duplicate removal keeps detections (> FG_THREAD) and the mean-teacher branch has pseudo labels
        sd = {k: v for k, v in sd.items() if not k.startswith("relation_nms.")}
        torch.nn.init.constant_(student.relation_nms.classifier.bias, 0.5)
    student.load_state_dict(sd, strict=False)
    teacher.load_state_dict(sd, strict=False)
    student.to(device)
    teacher.to(device)
    student.train()
    teacher.eval()
    opt = make_optimizer(cfg, student)
    sched = make_lr_scheduler(cfg, opt)
    loaders = {"source": [None] * cfg.SOLVER.MAX_ITER, "no_label": None}
    trainer = MTtrainer(student, teacher, loaders, opt, sched, None, None, 10 ** 9, cfg)
    init_teacher_weight(student, teacher)

    crop = CROP if crop is None else crop      # tests build the same trainer on small crops
    n_inst = N_INST if n_inst is None else n_inst
    n_lab = N_LAB if n_lab is None else n_lab           # tests: the cpu_baseline sample (1 + 1 crop) at full size
    n_unlab = N_UNLAB if n_unlab is None else n_unlab
    imgs, tgs = synthetic.make_labeled(n_lab, crop, n_inst, seed=1234 + rank)
    unl = synthetic.make_unlabeled(n_unlab, crop, cfg.MT.AUG_K + cfg.MT.AUG_S, seed=4321 + rank)
    targets = []
    for t in tgs:
        b = BoxList(t["boxes"].to(device), t["size"], "xyxy")
        b.add_field("labels", t["labels"].to(device))
        b.add_field("masks", SegmentationMask([[p for p in inst] for inst in t["polys"]], t["size"], mode="poly"))
        targets.append(b)
    imgs = imgs.to(device)
    unl = [u.to(device) for u in unl]

    def batch():
        # what the collators hand over (data/collate_batch.py:5-77): zero-padded to /32; hflip() mutates the
        # teacher ImageLists in place, so they are rebuilt from the resident crops every step
        return (to_image_list(list(imgs), cfg.DATALOADER.SIZE_DIVISIBILITY), targets,
                [to_image_list(list(u), cfg.DATALOADER.SIZE_DIVISIBILITY) for u in unl])

    return cfg, trainer, batch


def cpu_baseline():
    """The oracle (CPU restatement pinned to the reference, oracle/model.py::Trainer: the three forwards, loss weighting,
    backward, torch.optim.SGD with the reference's parameter groups, EMA) on a bounded sample -- full steps [A]-[E] on
    1 labeled + 1 unlabeled 1000x1000 crop = HALF a per-GPU batch -- 1 warm-up + 3 timed steps,
    median.  Baseline only."""
    import statistics
    import synthetic
    from oracle import model as om
    # intra-op threads are capped: on a 256-core host torch/oneDNN with 256 threads is ~20x SLOWER on these shapes
    # (measured: 775 s vs ~40 s); `cores` in the JSON is the thread count actually used
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    ss = json.load(open(os.path.join(ROOT, "tests", "golden", "state_shapes.json")))
    sd = synthetic.make_weights(ss["shapes"], seed=0)
    ot = om.Trainer(sd, om.default_cfg(), ss["trainable"], ss["param_order"])
    imgs, tgs = synthetic.make_labeled(1, CROP, N_INST, seed=1234)
    unl = synthetic.make_unlabeled(1, CROP, 3, seed=4321)
    targets = [om.Boxes(t["boxes"], t["size"], {"labels": t["labels"], "masks": t["polys"]}) for t in tgs]
    torch.manual_seed(0)
    times = []
    for i in range(4):
        t0 = time.time()
        ot.step(1400 + i, imgs, targets, unl)
        times.append(time.time() - t0)
    dt = statistics.median(times[1:])
    return {"value": round(2.0 / dt, 5), "unit": "imgs/sec", "cores": cores, "kind": "port",
            "sample": "full MT steps (sup fwd, teacher K=2xflip, student MGD+PSM, bwd, SGD, EMA) on 1 labeled + 1 "
                      "unlabeled 1000x1000 crop (half a per-GPU batch), fp32; 1 warm-up (%.1f s) + 3 timed steps, median "
                      "%.1f s (%s)" % (times[0], dt, ", ".join("%.1f" % t for t in times[1:]))}


NOMINAL_SCLK_MHZ = 2400.0  # the clock the dense matrix peaks are quoted at


class ClockSampler(object):
    """shader clock / board power of this process's GPU from its sysfs hwmon node, sampled every 10 ms by a helper thread
    (used in the bracketed profile leg only, never in the headline leg).  The MFMA peak scales with the clock, and under
    bf16 MFMA load this part runs at its board power limit well below the nominal clock (tools/clock_under_load.py:
    1.63 GHz and 1400 W under the dominant kernel back to back; 2.39 GHz under the fp32-input MFMA kernel)."""

    def __init__(self, device_index):
        import glob
        self.node = None
        try:
            bus = torch.cuda.get_device_properties(device_index).pci_bus_id
            dev = [d for d in glob.glob("/sys/bus/pci/devices/*") if d.lower().endswith(":%02x:00.0" % bus)]
            hw = glob.glob(dev[0] + "/hwmon/hwmon*/") if dev else []
            if hw and os.path.exists(hw[0] + "freq1_input"):
                self.node = hw[0]
        except Exception:
            self.node = None
        self.samples, self._stop, self._th = [], False, None

    def _read(self, name):
        try:
            return float(open(self.node + name).read())
        except Exception:
            return None

    def _run(self):
        pw = "power1_input" if os.path.exists(self.node + "power1_input") else "power1_average"
        while not self._stop:
            f, p = self._read("freq1_input"), self._read(pw)
            if f is not None:
                self.samples.append((f / 1e6, None if p is None else p / 1e6))
            time.sleep(0.01)

    def __enter__(self):
        if self.node is not None:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th is not None:
            self._th.join()

    def summary(self):
        if not self.samples:
            return None
        f = sorted(x[0] for x in self.samples)
        p = sorted(x[1] for x in self.samples if x[1] is not None)
        return {"sclk_mhz_median": round(f[len(f) // 2], 0), "sclk_mhz_p10_p90": [round(f[len(f) // 10], 0), round(f[len(f) * 9 // 10], 0)],
                "board_power_w_median": round(p[len(p) // 2], 0) if p else None, "samples": len(f)}


ARITH = {0: "fp32-input MFMA (IEEE fp32 products)",
         3: "fp32 tensors; products on the bf16 matrix pipe as a 3-term bf16 split (6 MFMAs, fp32 accumulate): error vs fp64 "
            "<= the fp32-input MFMA's (profiles/r01_precision.txt); MMT_CONV_PRECISION=0 selects the fp32-input MFMA",
         2: "fp32 tensors; 2-term bf16 split (3 MFMAs)", 1: "fp32 tensors; bf16 products, fp32 accumulate"}
ARITH_BF16_STORAGE = ("bf16 products, fp32 accumulate; activations and activation gradients of the ResNet body and the FPN "
                      "stored as bf16 (weights: bf16 planes of fp32 masters; heads, losses, weight gradients fp32)")


KERNEL_NAMES = {
    "fwd1": lambda mode: ("conv_fwd_kernel<128,128,2,2> (fp32 v_mfma_f32_32x32x2_f32)" if mode == 0 else
                          "conv_fwd_glds_kernel<128,128,4,1,%d,3> (v_mfma_f32_32x32x16_bf16 x %d products)" % (mode, PRODUCTS[mode])),
    "fwd5": lambda mode: "conv_pg_kernel<WM,KG,S> (plane-fed implicit GEMM, 64 WM x 128 tiles, 8 matrix + 4 copy waves, KG K groups per "
                         "block, v_mfma_f32_32x32x16_f16 x 3 products; brackets hold the kernel, the plane-split pass is bracketed apart)",
    "fwd4": lambda mode: "conv3x3_strip_kernel<TW,%d> (256x128 tiles on 8 waves, %s, v_mfma_f32_32x32x16_bf16 x %d "
                         "products; brackets hold the kernel and, in its split-K form, the finish launch)" % (
                             mode, "pre-split planes" if mode == 3 else "bf16 tensors as stored", PRODUCTS[mode]),
}


def csrc_sha1():
    """hash of the kernel sources (mmt-psm_amd/csrc/*.hip, *.h): ties profiles/pmc_traffic.json to the library it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mmt-psm_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def products_of(mode, dom):
    """matrix products per algorithmic multiply-add of a bracketed kernel group"""
    from maskrcnn_benchmark import _hip
    if mode == 3 and _hip.F16X2 and (dom in ("fwd4", "fwd5") or _hip.F16X2_TILED):
        return 3   # the two-term fp16 split: the strip kernel, and (F16X2_TILED) the tiled kernel too
    return PRODUCTS[mode]


def roofline(mode, ach, traffic, alg_bytes, flops, ms, prof, dt, nsteps, dom="fwd1"):
    """dominant kernel = the 128x128 forward tile (also runs every stride-1 data gradient).  `achieved` is algorithmic
    FLOP/s (2*M*N*K per launch / event-bracketed launch time); `peak` is the matrix-pipe peak available to that
    arithmetic: the fp32-input MFMA peak in mode 0, the bf16 dense peak divided by the products per multiply-add else."""
    n = max(len(prof), 1)
    nprod = products_of(mode, dom)
    peak = PEAK_FP32_MFMA_TFLOPS if mode == 0 else PEAK_BF16_MFMA_TFLOPS / nprod
    kern = KERNEL_NAMES[dom](mode)
    if nprod == 3 and dom == "fwd1":
        kern = "conv_fwd_glds_kernel<128,128,4,1,2,3,true> (activations split into two fp16 terms in registers, v_mfma_f32_32x32x16_f16 x 3 products)"
    if dom == "fwd5":
        kern = KERNEL_NAMES["fwd5"](mode)
    if nprod == 3 and dom == "fwd4":
        kern = ("conv3x3_strip_kernel<TW,2,true> (256x128 tiles on 8 waves, two fp16 planes per operand scaled per tensor, "
                "v_mfma_f32_32x32x16_f16 x 3 products; brackets hold the kernel and, in its split-K form, the finish launch)")
    r = {"bound": "mfma", "kernel": kern, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
         "frac": round(ach / peak, 4), "traffic": traffic,
         "algorithmic_bytes_per_launch": round(alg_bytes / n, 1), "algorithmic_flop_per_launch": round(flops / n, 1),
         "launches_per_step": len(prof) // max(nsteps, 1), "avg_launch_ms": round(ms / n, 4),
         "share_of_step_time": round(ms / (dt * 1e3), 4),
         "measured_in": "a separate leg of %d steps with an event pair around every launch of this kernel on its launch "
                        "stream (%.2f ms/step with the brackets); the headline leg carries no brackets" % (nsteps, dt / nsteps * 1e3)}
    if mode != 0:
        r["executed_mfma_tflops"] = round(ach * nprod, 1)
        r["peak_note"] = "%.0f TFLOP/s bf16 / fp16 dense / %d products; the same work on the fp32-input MFMA is capped at %.1f" % (
            PEAK_BF16_MFMA_TFLOPS, nprod, PEAK_FP32_MFMA_TFLOPS)
        r["vs_fp32_mfma_peak"] = round(ach / PEAK_FP32_MFMA_TFLOPS, 4)
    if nprod == 3:
        # the line the previous rounds were quoted on: the same algorithmic FLOP/s against the 6-product ceiling
        r["six_product_line"] = {"peak": round(PEAK_BF16_MFMA_TFLOPS / 6, 1), "frac": round(ach / (PEAK_BF16_MFMA_TFLOPS / 6), 4),
                                 "note": "3-term bf16 split (round-2 default, now the per-tensor fall-back; python bench.py --bf16x3)"}
    return r


HBM_ACHIEVABLE_TBS = 5.0   # what a copy kernel reaches on this part (profiles/r02_clock_under_load.txt: 4.9 - 5.5); the spec figure is 8


def conv_family(rec, steps, nprod):
    """every bracketed convolution / weight-gradient launch of `steps` single-stream steps (_hip.PROFILE with PROFILE_ALL) -> the
    FAMILY figure: sum over launches of the time a perfect kernel would need -- max(algorithmic FLOP at 2500 / nprod TFLOP/s,
    algorithmic bytes (input + weights + output + the fused epilogue's operands, each once) at 5 TB/s) -- against the sum of the
    measured launch times.  One number for the whole family: it cannot jump when another kernel becomes "dominant" (VERDICT r5
    weak 3).  Also per group (kind, kernel size, batch), as tools/conv_table.py prints them."""
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for r in rec:
        a = agg[r[3]]
        a[0] += 1
        a[1] += r[1].elapsed_time(r[2])
        a[2] = r[0]
        a[3] += r[6] if len(r) > 6 else 0.0
    rows = []
    for key, (c, ms, fl, eb) in agg.items():
        kind, N, Hh, W, Cin, Cout, KH, stride, ostride = key
        Ho, Wo = (Hh + stride - 1) // stride, (W + stride - 1) // stride
        if kind == "wgrad":
            byts = 4.0 * N * (Hh * W * Cin + Ho * Wo * Cout) + 4.0 * Cin * Cout * KH * KH
        else:
            byts = 4.0 * N * (Hh * W * Cin + Ho * Wo * Cout * (ostride if ostride else 1)) + 6.0 * Cin * Cout * KH * KH
        byts += eb / c
        t_m, t_h = fl / (PEAK_BF16_MFMA_TFLOPS * 1e12 / nprod) * 1e3, byts / (HBM_ACHIEVABLE_TBS * 1e12) * 1e3
        rows.append((key, c / steps, ms / c, fl, t_m, t_h))
    groups = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for key, c, per, fl, t_m, t_h in rows:
        k = ("wgrad " if key[0] == "wgrad" else "fwd/dgrad ") + ("fc" if key[2] == 1 else "%dx%d" % (key[6], key[6])) + (
            " N=%d" % key[1] if key[2] > 1 and key[1] <= 8 else "")
        g = groups[k]
        g[0] += c
        g[1] += c * per
        g[2] += c * max(t_m, t_h)
    time_ms = sum(c * per for _, c, per, _, _, _ in rows)
    bound_ms = sum(c * max(t_m, t_h) for _, c, _, _, t_m, t_h in rows)
    return rows, {"bound_ms": round(bound_ms, 3), "time_ms": round(time_ms, 3), "frac": round(bound_ms / time_ms, 4) if time_ms else None,
                  "launches_per_step": round(sum(c for _, c, _, _, _, _ in rows), 1),
                  "algorithmic_tflop_per_step": round(sum(c * fl for _, c, _, fl, _, _ in rows) / 1e12, 3),
                  "groups": {k: {"launches": round(v[0], 1), "time_ms": round(v[1], 3), "bound_ms": round(v[2], 3),
                                 "frac": round(v[2] / v[1], 3) if v[1] else None} for k, v in sorted(groups.items(), key=lambda kv: -kv[1][1])},
                  "definition": "sum over every convolution / weight-gradient launch of a step of max(2 M N K / (%.0f / %d TFLOP/s), "
                                "(input + weights + output + fused-epilogue operands) / %.0f TB/s) divided by the sum of their "
                                "event-bracketed durations, single stream (teacher on the step stream, weight gradients on the step "
                                "stream), %d steps; the same table as mmt-psm_amd/tools/conv_table.py" % (
                                    PEAK_BF16_MFMA_TFLOPS, nprod, HBM_ACHIEVABLE_TBS, steps)}


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: re-execute this command line under torch.distributed.run with
    N ranks on this node (127.0.0.1 rendezvous on a free port).  Fails -- non-zero, before anything is timed -- when the node
    has fewer than N GPUs: a multi-GPU line must never be produced by fewer ranks than it names."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; one process per GPU, no "
                         "oversubscription -- not running" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank run failed (torch.distributed.run exit code %d)" % (n, rc))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # 10 warm-up + 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--profile-steps", type=int, default=10, help="steps of the separate event-bracketed leg (roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--irnet", action="store_true", help="IR-Net on (BASELINE configs[4] in fp32); not the headline line")
    ap.add_argument("--f16x2", action="store_true", help="(default since round 3; kept for old command lines) convolutions on the "
                    "two-term fp16 split: 3 matrix products per multiply instead of 6")
    ap.add_argument("--bf16x3", action="store_true", help="the round-2 default arithmetic, now the per-tensor fall-back: 3-term bf16 "
                    "split, 6 matrix products per multiply (MMT_F16X2=0)")
    ap.add_argument("--supervised", action="store_true", help="BASELINE configs[1]: Mask R-CNN R50-FPN supervised-only (MT.LAMBDA 0), "
                    "bs = 4 labeled 1000x1000 crops per GPU, fwd + bwd + SGD; not the headline line")
    ap.add_argument("--bf16", action="store_true", help="BASELINE configs[4]'s arithmetic: bf16 products (fp32 accumulate) and bf16 "
                    "activation storage in the backbone + FPN (MMT_CONV_PRECISION=1 MMT_BF16_STORAGE=1); not the headline line")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` outside a launcher: start the N ranks here (one process per GPU, RCCL over xGMI) --
        # the line this process would otherwise print is a one-rank number labelled with a flag nobody read
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks -- refusing to print a line whose "
                         "n_gpus does not match the request" % (args.gpus, world))
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own: %d device(s) visible, one process per GPU "
                         "is the only supported layout" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("MMT_FORCE_DIST") == "1"  # the latter: exercise RCCL on a 1-GPU box
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    from maskrcnn_benchmark import _hip
    _hip.lib()
    if args.bf16:
        _hip.set_conv_precision(1)
        _hip.set_bf16_storage(True)
    if args.f16x2:
        _hip.set_f16x2(True)
    if args.bf16x3:
        _hip.set_f16x2(False)
    cfg, trainer, batch = build(device, rank, args.irnet, base_lr=BENCH_BASE_LR, supervised=args.supervised,
                                n_lab=4 if args.supervised else None)
    per_gpu = 4 if args.supervised else N_LAB + N_UNLAB
    it0 = cfg.MT.START_MT + cfg.MT.RAMPUP_STEP + 100  # mean-teacher branch active, consistency weight = lambda

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def step(i):
        il, targets, ul = batch()
        return trainer.train_step(it0 + i, il, targets, None if args.supervised else ul)

    def timed(first, n, profile):
        """n steps between two (barrier + synchronize) pairs.  profile=False: the headline leg -- nothing but one event per
        step boundary on the step stream (per-step durations for the median, no host sync).  profile=True: the separate
        leg in which every launch of the dominant kernel is bracketed by an event pair on its launch stream."""
        sync()
        _hip.PROFILE = [] if profile else None
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        c0 = _hip.C_CALLS[0]
        t0 = time.perf_counter()
        for i in range(n):
            marks[i].record()
            losses = step(first + i)
        marks[n].record()
        sync()
        dt = time.perf_counter() - t0
        timed.c_calls_per_step = (_hip.C_CALLS[0] - c0) / max(n, 1)
        # the two large-tile forward / data-gradient kernels: 'fwd4' = conv3x3_strip_kernel (its brackets include the
        # plane-split pass of its input), 'fwd1' = conv_fwd_glds_kernel<128,128>; the roofline line is the one with more time
        groups = {}
        for q in (_hip.PROFILE or []):
            groups.setdefault(q[3][0], []).append(q)
        _hip.PROFILE = None
        prof = groups
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof, losses, per_step

    mode = _hip.get_conv_precision()
    trace_env = os.environ.get("MMT_DIST_TRACE") == "1"
    for i in range(args.warmup):
        step(i)
    skipped0 = trainer.skipped_pairs
    dt, _, losses, per_step = timed(args.warmup, args.steps, False)
    c_calls = timed.c_calls_per_step
    skipped = trainer.skipped_pairs - skipped0
    if use_dist:
        t = torch.tensor([skipped], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        skipped = int(t.item())
    nxt = args.warmup + args.steps
    npf = max(1, min(args.profile_steps, args.steps))
    if use_dist:   # the bracketed leg also records when each piece of the gradient went out and arrived (`dist_trace`):
        os.environ["MMT_DIST_TRACE"] = "1"   # the trace synchronises per step, so it is off in the headline leg
    with ClockSampler(local) as clock:
        dtp, groups, _, _ = timed(nxt, npf, True)
    if use_dist and not trace_env:
        os.environ.pop("MMT_DIST_TRACE", None)
    nxt += npf

    def total(g):
        return sum(q[0] for q in g), sum(q[1].elapsed_time(q[2]) for q in g)

    # dominant kernel = the large-tile forward kernel with more bracketed time in this leg
    ranked = sorted(("fwd1", "fwd4", "fwd5"), key=lambda k: -total(groups.get(k, []))[1])
    dom, odom_ = ranked[0], ranked[1]
    prof = groups.get(dom, [])
    other = groups.get(odom_, [])
    single = None
    if world == 1 and trainer.overlap_teacher and not os.environ.get("MMT_BENCH_NO_FP32_LEG"):
        # the same launches with the teacher on the main stream: kernel durations without a second stream sharing the GPU
        trainer.overlap_teacher = False
        step(nxt)
        n1 = max(2, npf // 2)
        dt1, g1, _, _ = timed(nxt + 1, n1, True)
        nxt += 1 + n1
        fl1, ms1 = total(g1.get(dom, []))
        trainer.overlap_teacher = True
        if ms1 > 0:
            single = {"ms_per_step": round(dt1 / n1 * 1e3, 3), "steps": n1, "achieved": round(fl1 / (ms1 * 1e-3) / 1e12, 2)}
    family = fwd_leg = None
    if world == 1 and mode == 3 and not args.supervised and not os.environ.get("MMT_BENCH_NO_FAMILY_LEG"):
        # (i) the convolution FAMILY against its bound: every conv / weight-gradient launch of three single-stream steps bracketed
        ov = trainer.overlap_teacher
        trainer.overlap_teacher = False
        try:
            step(nxt)
            sync()
            _hip.PROFILE, _hip.PROFILE_ALL = [], True
            nf = 3
            for i in range(nf):
                step(nxt + 1 + i)
            sync()
            rec, _hip.PROFILE, _hip.PROFILE_ALL = _hip.PROFILE, None, False
            nxt += 1 + nf
            _, family = conv_family(rec, nf, products_of(mode, "fwd4"))
            # (ii) the FORWARD leg north_star quotes its target on: teacher forward_teacher + the student's supervised and
            # unsupervised forwards without autograd, one stream; algorithmic FLOP = sum of 2 M N K over its convolution / fc
            # launches (counted in one bracketed pass), time from un-bracketed passes
            il, targets, ul = batch()
            trainer.forward_only(il, targets, ul)
            sync()
            _hip.PROFILE, _hip.PROFILE_ALL = [], True
            il, targets, ul = batch()
            trainer.forward_only(il, targets, ul)
            sync()
            frec, _hip.PROFILE, _hip.PROFILE_ALL = _hip.PROFILE, None, False
            fl_fwd = sum(q[0] for q in frec)
            nfw = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync()
            t0 = time.perf_counter()
            e0.record()
            for i in range(nfw):
                il, targets, ul = batch()
                trainer.forward_only(il, targets, ul)
            e1.record()
            sync()
            wall = (time.perf_counter() - t0) / nfw
            dev = e0.elapsed_time(e1) / nfw * 1e-3
            _, ffam = conv_family(frec, 1, products_of(mode, "fwd4"))
            peak3 = PEAK_BF16_MFMA_TFLOPS / products_of(mode, "fwd4")
            fwd_leg = {"what": "teacher forward_teacher (K-aug x flip, 8 views) + student supervised forward + student unsupervised "
                               "forward with their losses, torch.no_grad, one stream: 12 image-forwards of 1000x1000 per pass "
                               "(the coarse inference reuses view 0's pyramid)",
                       "passes": nfw, "ms_per_pass": round(wall * 1e3, 3), "device_ms_per_pass": round(dev * 1e3, 3),
                       "algorithmic_tflop_per_pass": round(fl_fwd / 1e12, 3), "conv_launches_per_pass": len(frec),
                       "achieved_tflops": round(fl_fwd / wall / 1e12, 2),
                       "frac_of_833": round(fl_fwd / wall / 1e12 / peak3, 4),
                       "frac_of_fp32_mfma_peak_157": round(fl_fwd / wall / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                       "conv_family_of_this_leg": {k: ffam[k] for k in ("bound_ms", "time_ms", "frac", "launches_per_step")},
                       "note": "north_star: >= 0.5 x MFMA roofline on the student+teacher forward.  Against the fp32-input MFMA peak "
                               "(157.3 TFLOP/s: the roofline of an fp32 network on this part) the leg is far above 0.5; against the "
                               "fp16 pipe at 3 products per multiply (833) it is what frac_of_833 says; the leg's own bound -- many of "
                               "its launches are HBM-bound 1x1 layers -- is conv_family_of_this_leg.bound_ms"}
        finally:
            trainer.overlap_teacher = ov
            _hip.PROFILE, _hip.PROFILE_ALL = None, False
    ref_fp32 = None
    if mode != 0 and world == 1 and not os.environ.get("MMT_BENCH_NO_FP32_LEG"):
        # the same workload on the fp32-input MFMA kernels (mode 0), reported next to the headline number
        _hip.set_conv_precision(0)
        step(nxt)
        n0 = max(2, npf // 2)
        dt0, _, _, _ = timed(nxt + 1, n0, False)
        _, g0, _, _ = timed(nxt + 1 + n0, n0, True)
        fl0, ms0 = total(g0.get("fwd1", []))
        ref_fp32 = {"ms_per_step": round(dt0 / n0 * 1e3, 3), "value": round(per_gpu * n0 / dt0, 4),
                    "steps": n0, "dominant_kernel_tflops": round(fl0 / (ms0 * 1e-3) / 1e12, 2) if ms0 > 0 else None,
                    "frac_of_fp32_mfma_peak": round(fl0 / (ms0 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ms0 > 0 else None}
        _hip.set_conv_precision(mode)

    if rank == 0:
        imgs_per_step = per_gpu * world
        value = imgs_per_step * args.steps / dt          # whole-job throughput over the K bracketed steps
        import statistics
        med = statistics.median(per_step)                 # per-step durations from the step-boundary events (rank 0)
        flops = sum(p[0] for p in prof)
        ms = sum(p[1].elapsed_time(p[2]) for p in prof)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # algorithmic HBM bytes of those launches: input + weights + output once
        alg_bytes = sum(4.0 * (k[1] * k[2] * k[3] * k[4] + k[5] * k[4] * k[6] * k[6]
                               + k[1] * (k[2] // k[7]) * (k[3] // k[7]) * k[5]) for k in (p[3] for p in prof))
        # ... plus what the fused epilogues of those launches read (residual / top-down add, ReLU mask of a data gradient):
        # operands of the launch like its input, read once (round 3 left them out and called the difference waste)
        alg_bytes += sum(p[6] for p in prof if len(p) > 6)
        traffic, traffic_note = None, None
        try:  # HBM/fabric bytes per launch of this kernel from the committed PMC passes (not measurable live) -- quoted only while
            # the kernels are the ones the counters were collected on: the file carries the hash of csrc/ 
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if tj.get("csrc_sha1") == csrc_sha1():
                traffic = tj["by_mode"][str(mode)].get("traffic_bytes_per_launch_" + dom, tj["by_mode"][str(mode)].get(
                    "traffic_bytes_per_launch") if dom == "fwd1" else None)
            else:
                traffic_note = ("profiles/pmc_traffic.json was collected on other kernel sources (csrc hash %s, now %s): re-run "
                                "mmt-psm_amd/tools/make_profiles.sh" % (str(tj.get("csrc_sha1"))[:12], csrc_sha1()[:12]))
        except Exception:
            pass
        out = {
            "metric": "imgs/sec (student+teacher step, 1000x1000, AUG_K=2)",
            "value": round(value, 4), "unit": "imgs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "median_ms_per_step": round(med, 3),
            "p10_p90_ms_per_step": [round(sorted(per_step)[len(per_step) // 10], 3),
                                    round(sorted(per_step)[(len(per_step) * 9) // 10 - (1 if len(per_step) >= 10 else 0)], 3)],
            "imgs_per_sec_at_median": round(imgs_per_step / (med * 1e-3), 4),
            "higher_is_better": True, "scaling": "weak",
            "library_calls_per_step": round(c_calls, 1),   # interpreter -> C-ABI calls of both launch-issuing threads
            "rccl_ranks": dist.get_world_size() if use_dist else 0,
            "backend": (dist.get_backend() + " (RCCL)") if use_dist else "none (single process, no process group)",
            "vs_baseline": None, "dtype": "bf16" if mode == 1 else "f32", "data": "synthetic",
            "config": {"workload": ("Mask R-CNN R50-FPN supervised-only step (BASELINE configs[1]): per GPU 4 labeled 1000x1000x3 "
                                    "crops, MT.LAMBDA 0, fwd+bwd+SGD, fp32, IR-Net %s" if args.supervised else
                                    "MMT-PSM mean-teacher step (BASELINE configs[2]; configs[3] when n_gpus>1): per GPU 2 "
                                    "labeled + 2 unlabeled 1000x1000x3 crops, AUG_K=2 + flip, AUG_S=1, MT.LAMBDA 5, "
                                    "PSM+MGD, EMA teacher, fwd+bwd+SGD, R50-FPN fp32, IR-Net %s") % (
                                       "ON (relation NMS + mask relation; RELATION_NMS.LOSS 0.01)" if args.irnet else "off"),
                       "image_forwards_per_step_per_gpu": 4 if args.supervised else 12, "parallelism": "dp%d" % world,
                       "base_lr": BENCH_BASE_LR, "consistency_branch_skipped_steps": skipped,
                       "losses": {k: round(float(v.detach()), 5) for k, v in losses.items()}},
            "roofline": dict(roofline(mode, ach, traffic, alg_bytes, flops, ms, prof, dtp, npf, dom),
                             **({"traffic_note": traffic_note} if traffic_note else {})),
        }
        pres = [q[5] for q in prof if len(q) > 5 and q[5] is not None]
        if pres:
            # launches whose input planes no producing epilogue had written: one split pass over the input in front of them
            ps = sum(a_.elapsed_time(b_) for a_, b_ in pres)
            out["roofline"]["plane_split_pass"] = {
                "launches_per_step": len(pres) // npf, "ms_per_step": round(ps / npf, 4),
                "achieved_with_it": round(flops / ((ms + ps) * 1e-3) / 1e12, 2),
                "frac_with_it": round(flops / ((ms + ps) * 1e-3) / 1e12 / out["roofline"]["peak"], 4)}
        ck = clock.summary()
        if ck is not None and mode != 0:
            # what the part clocked at during the bracketed leg (whole step, all kernels): the matrix peak at that clock
            ck["peak_at_median_sclk"] = round(out["roofline"]["peak"] * ck["sclk_mhz_median"] / NOMINAL_SCLK_MHZ, 1)
            ck["frac_of_peak_at_median_sclk"] = round(out["roofline"]["achieved"] / ck["peak_at_median_sclk"], 4)
            ck["note"] = ("sysfs hwmon of this GPU, 10 ms samples over the bracketed leg; `peak` above is quoted at %d MHz, under "
                          "bf16 / fp16 MFMA load the part sits at its board power limit below that (profiles/r03_clock_under_load.txt)"
                          % NOMINAL_SCLK_MHZ)
        if ck is not None:
            out["roofline"]["clock"] = ck
        if other:
            fo, mo = total(other)
            odom = odom_
            opeak = PEAK_FP32_MFMA_TFLOPS if mode == 0 else PEAK_BF16_MFMA_TFLOPS / products_of(mode, odom)
            out["roofline"]["other_large_tile_kernel"] = {
                "kernel": (("conv_fwd_glds_kernel<128,128,4,1,2,3,true> (raw fp32 rows split into two fp16 terms in registers, "
                            "v_mfma_f32_32x32x16_f16 x 3 products)" if odom == "fwd1" else
                            KERNEL_NAMES["fwd5"](mode) if odom == "fwd5" else
                            "conv3x3_strip_kernel<TW,2,true> (two fp16 planes per operand, v_mfma_f32_32x32x16_f16 x 3 products)")
                           if products_of(mode, odom) == 3 else KERNEL_NAMES[odom](mode)),
                "launches_per_step": len(other) // npf,
                "achieved": round(fo / (mo * 1e-3) / 1e12, 2), "frac": round(fo / (mo * 1e-3) / 1e12 / opeak, 4),
                "avg_launch_ms": round(mo / len(other), 4), "share_of_step_time": round(mo / (dtp * 1e3), 4)}
            if mode != 0:   # the 6-product line of rounds 1-2, as for the dominant kernel
                out["roofline"]["other_large_tile_kernel"]["six_product_frac"] = round(
                    fo / (mo * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 6), 4)
        out["roofline"]["step_bound"] = (
            "the device: ~49 ms of kernel time per step over three streams at ~1.45 x overlap; the host runs ahead of the device except "
            "for ~2.5 ms between the teacher's last result and the consistency backward (profiles/r04_host_device_phases.txt).  What "
            "moved the step in round 4 was device work removed (row-resident 1x1 kernel's register epilogue -1.7 ms) and a stream "
            "priority (-1.6 ms under a process group); host work removed, chains shortened, launches folded, ring depths, split-K "
            "forms and block counts re-tuned did not (DESIGN.md section 5, profiles/r04_history.md); the kernels' in-step durations "
            "are read while up to three streams share the GPU")
        # the launches of that kernel with an un-split K (the large FPN / layer1-2 shapes); the others are the few-tile,
        # long-K layers, whose bracketed duration also contains their small finish launch
        uns = [q for q in prof if len(q) < 5 or q[4] <= 1]
        if uns and len(uns) < len(prof):
            fu, mu = sum(q[0] for q in uns), sum(q[1].elapsed_time(q[2]) for q in uns)
            out["roofline"]["unsplit_k_launches"] = {
                "launches_per_step": len(uns) // max(npf, 1), "achieved": round(fu / (mu * 1e-3) / 1e12, 2),
                "frac": round(fu / (mu * 1e-3) / 1e12 / out["roofline"]["peak"], 4), "avg_launch_ms": round(mu / len(uns), 4)}
        out["config"]["conv_arithmetic"] = ARITH_BF16_STORAGE if _hip.bf16_storage() else ARITH[mode]
        if _hip.F16X2 and mode == 3:
            out["config"]["conv_arithmetic"] = (
                "fp32 tensors; products on the fp16 matrix pipe as a two-term fp16 split of both operands, each tensor scaled by a "
                "power of two derived on the device from its largest magnitude (3 MFMAs per multiply, fp32 accumulate): error vs "
                "fp64 <= the 3-term bf16 split's and the fp32-input MFMA's (profiles/r03_precision_f16x2.txt); tensors whose crest "
                "factor max/mean exceeds 2^17 fall back to the 3-term bf16 split (6 MFMAs) per consuming site; "
                "MMT_F16X2=0 / --bf16x3 selects that split everywhere, MMT_CONV_PRECISION=0 the fp32-input MFMA")
            out["config"]["f16_split_launches"] = dict(_hip.F16_STATS)
        if single is not None:
            # `achieved` above is bracketed on the launch stream while the teacher's stream shares the GPU; this is the same
            # kernel, same shapes, in a single-stream run of the same step (MMT_OVERLAP_TEACHER=0)
            single["frac"] = round(single["achieved"] / out["roofline"]["peak"], 4)
            out["roofline"]["single_stream"] = single
        if family is not None:
            out["roofline"]["family"] = family
        if fwd_leg is not None:
            out["forward_leg"] = fwd_leg
        if ref_fp32 is not None:
            out["fp32_mfma_mode"] = ref_fp32
        if use_dist and getattr(trainer._bucketed, "last_trace", None) is not None:
            # MMT_DIST_TRACE=1: when each piece of the flat gradient went out and arrived, on the device clock of the last step
            out["dist_trace"] = dict(trainer._bucketed.last_trace, backend=dist.get_backend(), world=world,
                                     teacher_stream_priority=trainer.t_stream.priority if trainer.t_stream is not None else None,
                                     grad_mbytes=round(trainer.flat_s.grad.numel() * 4 / 1e6, 1))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
