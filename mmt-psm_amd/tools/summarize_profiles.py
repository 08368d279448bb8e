"""gpurun_out/prof/* (make_profiles.sh) -> profiles/r02_* + profiles/pmc_traffic.json + profiles/r02_summary.md"""
import csv, json, os, shutil
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
SRC, DST, TAG = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles"), "r02_"
KERN = {3: {"fwd4": "conv3x3_strip_kernel<", "fwd1": "conv_fwd_glds_kernel<128, 128, 4, 1, 3, 3>"},
        0: {"fwd1": "conv_fwd_kernel<128, 128, 2, 2>"}}


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "at::").split("(")[0][:70]


out = ["# Round 2 -- profiles of `python bench.py` on 1 x MI355X (state at the end of the round)", "",
       "Produced by `mmt-psm_amd/tools/make_profiles.sh` (GPU box) + `mmt-psm_amd/tools/summarize_profiles.py`. Arithmetic modes: "
       "**mode 3** = default (3-term bf16 split on the bf16 matrix pipe, fp32-grade), **mode 0** = fp32-input MFMA "
       "(`MMT_CONV_PRECISION=0`). Files: `r02_bench_default.json` (un-profiled `python bench.py`: 10 warm-up + 50 timed steps, "
       "median next to the mean, event brackets in a separate 10-step leg, CPU baseline 1 + 3 steps); per mode "
       "`r02_kernel_stats_modeM.csv` (rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --profile-steps 5 "
       "--no-cpu-baseline`), `r02_bench_under_rocprof_modeM.json` (the line that run printed), "
       "`r02_pmc_{FETCH,WRITE}_SIZE_by_kernel_modeM.csv` (two separate --pmc passes, --kernel-trace only); "
       "`r02_pmc_mfma_busy.txt` (one SQ pass, single-stream); `pmc_traffic.json` = what bench.py reports as roofline.traffic.", ""]
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of `python bench.py "
                     "--steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline`; per-kernel tables profiles/r02_pmc_*_by_kernel_mode*.csv",
           "fetch_correction": 2.0,
           "note": "FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): "
                   "doubled. WRITE_SIZE uncalibrated, taken as is. Infinity-Cache hits are counted, so this is fabric traffic, an "
                   "upper bound on HBM traffic.", "by_mode": {}}
bd = json.load(open(os.path.join(SRC, "bench_default.json")))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, TAG + "bench_default.json"))
shutil.copy(os.path.join(SRC, "pmc_mfma_busy.txt"), os.path.join(DST, TAG + "pmc_mfma_busy.txt"))
r = bd["roofline"]
o = r.get("other_large_tile_kernel", {})
out += ["Un-profiled default run: **%.1f imgs/s, %.2f ms/step** (mean of 50 bracketed steps; median %.2f ms, p10/p90 %s). "
        "Dominant kernel `%s`: %.1f TFLOP/s algorithmic over %d launches/step (frac %.3f of 2500/6 = 417; %.1f %% of step time; "
        "single-stream leg %.1f TFLOP/s = %.3f); the other large-tile kernel `%s`: %.1f TFLOP/s over %d launches/step. Same step in "
        "mode 0: %.1f imgs/s, %.1f ms/step, dominant kernel %.1f TFLOP/s (frac %.3f of 157.3). CPU baseline (oracle, %d threads): "
        "%.3f imgs/s." % (
            bd["value"], bd["ms_per_step"], bd["median_ms_per_step"], bd["p10_p90_ms_per_step"], r["kernel"].split(" ")[0],
            r["achieved"], r["launches_per_step"], r["frac"], 100 * r["share_of_step_time"], r["single_stream"]["achieved"],
            r["single_stream"]["frac"], o.get("kernel", "-").split(" ")[0], o.get("achieved", 0), o.get("launches_per_step", 0),
            bd["fp32_mfma_mode"]["value"], bd["fp32_mfma_mode"]["ms_per_step"], bd["fp32_mfma_mode"]["dominant_kernel_tflops"],
            bd["fp32_mfma_mode"]["frac_of_fp32_mfma_peak"], bd["cpu_baseline"]["cores"], bd["cpu_baseline"]["value"]), ""]
for mode in (3, 0):
    rows = list(csv.DictReader(open(os.path.join(SRC, "kernel_stats_mode%d.csv" % mode))))
    for f in ("kernel_stats_mode%d.csv", "bench_under_rocprof_mode%d.json", "pmc_FETCH_SIZE_by_kernel_mode%d.csv",
              "pmc_WRITE_SIZE_by_kernel_mode%d.csv"):
        shutil.copy(os.path.join(SRC, f % mode), os.path.join(DST, TAG + f % mode))
    b = json.load(open(os.path.join(SRC, "bench_under_rocprof_mode%d.json" % mode)))
    nsteps = b["steps"] + b["warmup"] + 5 + (1 + 2 if mode == 3 else 0)   # timed + warm-up + profile leg (+ single-stream leg)
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    nl = sum(int(x["Calls"]) for x in rows)
    out += ["## mode %d" % mode, "",
            "bench line under the profiler: %.2f imgs/s, %.1f ms/step; roofline (`%s`): %.1f TFLOP/s (frac %.3f), avg launch %.4f ms"
            % (b["value"], b["ms_per_step"], b["roofline"]["kernel"].split(" ")[0], b["roofline"]["achieved"], b["roofline"]["frac"],
               b["roofline"]["avg_launch_ms"]), "",
            "%d steps traced: %.1f ms of kernel time = %.1f ms/step (sum over both streams), %d launches/step" % (
                nsteps, tot / 1e6, tot / 1e6 / nsteps, nl // nsteps), "",
            "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for x in rows[:18]:
        out.append("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(x["Name"]), x["Calls"], float(x["TotalDurationNs"]) / 1e6,
                                                          float(x["AverageNs"]) / 1e3, float(x["Percentage"])))
    fe = list(csv.DictReader(open(os.path.join(SRC, "pmc_FETCH_SIZE_by_kernel_mode%d.csv" % mode))))
    wr = list(csv.DictReader(open(os.path.join(SRC, "pmc_WRITE_SIZE_by_kernel_mode%d.csv" % mode))))
    tm = traffic["by_mode"].setdefault(str(mode), {})
    for key, name in KERN[mode].items():
        f1 = [x for x in fe if name in x["kernel"]]
        w1 = [x for x in wr if name in x["kernel"]]
        k1 = [x for x in rows if name in x["Name"]]
        if not (f1 and w1 and k1):
            continue
        # (both strip widths of conv3x3_strip_kernel are one entry, as in bench.py's 'fwd4' group)
        nd = sum(int(x["dispatches"]) for x in f1)
        fkb = sum(float(x["FETCH_SIZE_sum"]) for x in f1) / nd
        wkb = sum(float(x["WRITE_SIZE_sum"]) for x in w1) / max(sum(int(x["dispatches"]) for x in w1), 1)
        avg_us = sum(float(x["TotalDurationNs"]) for x in k1) / sum(int(x["Calls"]) for x in k1) / 1e3
        tb = (2.0 * fkb + wkb) * 1024
        tm["traffic_bytes_per_launch_" + key] = tb
        tm["kernel_" + key] = name
        tm["dispatches_" + key] = nd
        live = None
        rr = b["roofline"]
        if name.split("<")[0] in rr["kernel"]:
            live, alg = rr["avg_launch_ms"] * 1e3, rr["algorithmic_bytes_per_launch"]
        elif "other_large_tile_kernel" in rr and name.split("<")[0] in rr["other_large_tile_kernel"]["kernel"]:
            live, alg = rr["other_large_tile_kernel"]["avg_launch_ms"] * 1e3, None
        out += ["", "`%s`: rocprof average %.1f us per launch%s. PMC per launch (%s dispatches): FETCH_SIZE %.0f KB (x2 gfx950 correction = "
                "%.1f MB), WRITE_SIZE %.0f KB -> fabric traffic %.1f MB%s." % (
                    name + ("...>" if name.endswith("<") else ""), avg_us,
                    (" vs %.1f us measured live by bench.py with events on the launch stream, same command%s" % (
                        live, " (its brackets also hold the split-K finish launch of the call)")) if live else "",
                    nd, fkb, 2 * fkb * 1024 / 1e6, wkb, tb / 1e6,
                    (" vs %.1f MB algorithmic (input + weights + output once)" % (alg / 1e6)) if live and alg else "")]
    if mode == 3 and "traffic_bytes_per_launch_fwd1" in tm:
        tm["traffic_bytes_per_launch"] = tm["traffic_bytes_per_launch_fwd1"]
    out.append("")
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
# how the step is put together + the other configurations (text files of tools/, copied as they are)
extra = [("conv_table.txt", "every convolution call of a step by shape (single-stream, event-bracketed): time, TFLOP/s, MFMA / HBM bound"),
         ("host_device_phases.txt", "host issue time and device arrival time of every phase of one overlapped step (no profiler)"),
         ("stream_timeline.txt", "occupancy of the two streams per ms of one step (rocprofv3 kernel trace; slower than un-profiled)"),
         ("step_series_recipe_lr.txt", "per-step ms over 120 steps at the RECIPE's learning rate (* = consistency branch skipped)"),
         ("step_series_bench.txt", "the same with the bench's frozen learning rate"),
         ("clock_under_load.txt", "shader clock and board power of the GPU under the dominant kernel back to back, the fp32-input MFMA kernel, an HBM copy, the bench step"),
         ("bench_f16x2.json", "`python bench.py --f16x2`: the strip kernel's 3x3 convolutions on the two-term fp16 split (3 products), everything else as the default"),
         ("precision_f16x2.txt", "time and error against fp64 of that arithmetic, the default 3-term bf16 split and the fp32-input MFMA on the strip shapes, for activation-like, gradient-like, extreme-scale and outlier-laden operands"),
         ("bench_bf16.json", "`python bench.py --bf16`: bf16 products + bf16 activation storage"),
         ("bench_bf16_irnet.json", "`python bench.py --bf16 --irnet` = BASELINE configs[4] on one GPU"),
         ("bench_irnet.json", "`python bench.py --irnet`: IR-Net on, fp32-grade arithmetic")]
out += ["## How the step is put together; other configurations", ""]
for f, what in extra:
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, TAG + f))
        line = "* `profiles/%s%s` -- %s" % (TAG, f, what)
        if f.endswith(".json"):
            try:
                j = json.loads(open(os.path.join(SRC, f)).read().strip().splitlines()[-1])
                line += ": **%.1f imgs/s, %.2f ms/step** (median %.2f; skipped-branch steps %s)" % (
                    j["value"], j["ms_per_step"], j["median_ms_per_step"], j["config"].get("consistency_branch_skipped_steps"))
            except Exception as e:
                line += " (unreadable: %s)" % e
        out.append(line)
out.append("")
out += ["## MFMA-busy (single-stream SQ pass)", "", "```"] + [l.rstrip() for l in open(os.path.join(SRC, "pmc_mfma_busy.txt")) if "mfma_busy_fraction" in l or l.startswith("#")] + ["```", ""]
hist = open(os.path.join(DST, "r02_history.md")).read() if os.path.exists(os.path.join(DST, "r02_history.md")) else ""
open(os.path.join(DST, "r02_summary.md"), "w").write("\n".join(out) + "\n" + hist)
print("\n".join(out)[:4000])
