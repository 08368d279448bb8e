"""gpurun_out/prof/* (make_profiles.sh) -> profiles/r06_* + profiles/pmc_traffic.json + profiles/r06_summary.md.
The number of steps a kernel trace holds is COUNTED (one `ema_kernel` launch per step), not assumed."""
import csv, json, os, shutil, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
SRC, DST, TAG = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles"), "r06_"
ARITH = {"f16x2": "default: two-term fp16 split, 3 products per multiply (peak 2500 / 3 = 833 TFLOP/s)",
         "bf16x3": "`MMT_F16X2=0` / `--bf16x3`: three-term bf16 split, 6 products (round-2 default, now the per-tensor fall-back; peak 417)",
         "mode0": "`MMT_CONV_PRECISION=0`: fp32-input MFMA (peak 157.3)"}
KERN = {"f16x2": {"fwd4": "conv3x3_strip_kernel<", "fwd1": "conv_fwd_glds_kernel<128, 128, 4, 1, 2, 3, true>", "fwd5": "conv_pg_kernel<"},
        "bf16x3": {"fwd4": "conv3x3_strip_kernel<", "fwd1": "conv_fwd_glds_kernel<128, 128, 4, 1, 3, 3, false>"},
        "mode0": {"fwd1": "conv_fwd_kernel<128, 128, 2, 2>"}}
MODE_OF = {"f16x2": "3", "bf16x3": "3_bf16x3", "mode0": "0"}


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "at::").split("(")[0][:72]


def is_library(n):
    return n.startswith("void at::") or n.startswith("at::") or "rocprim" in n or "rocclr" in n or "hipcub" in n


out = ["# Round 6 -- profiles of `python bench.py` on 1 x MI355X (state at the end of the round)", "",
       "Produced by `mmt-psm_amd/tools/make_profiles.sh` (GPU box) + `mmt-psm_amd/tools/summarize_profiles.py`. Files: "
       "`r06_bench_default.json` (un-profiled `python bench.py`: 10 warm-up + 50 timed steps, median next to the mean, event brackets "
       "in a separate 10-step leg, CPU baseline 1 + 3 steps); per arithmetic `r06_kernel_stats_<tag>.csv` (rocprofv3 --kernel-trace "
       "--stats of `bench.py --steps 5 --warmup 2 --profile-steps 5 --no-cpu-baseline`), `r06_bench_under_rocprof_<tag>.json` (the line "
       "that run printed), `r06_pmc_{FETCH,WRITE}_SIZE_by_kernel_<tag>.csv` (two separate --pmc passes, --kernel-trace only); "
       "`r06_pmc_mfma_busy.txt` (one SQ pass, single-stream); `pmc_traffic.json` = what bench.py reports as roofline.traffic. "
       "Tags: " + "; ".join("**%s** = %s" % kv for kv in ARITH.items()) + ".", ""]
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of `python bench.py "
                     "--steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline`; per-kernel tables profiles/r06_pmc_*_by_kernel_*.csv",
           "fetch_correction": 2.0, "csrc_sha1": __import__("importlib").import_module("bench").csrc_sha1(),
           "note": "FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): "
                   "doubled. WRITE_SIZE uncalibrated, taken as is. Infinity-Cache hits are counted, so this is fabric traffic, an "
                   "upper bound on HBM traffic.  by_mode['3'] = the default arithmetic of mode 3 (two-term fp16 split).", "by_mode": {}}
bd = json.load(open(os.path.join(SRC, "bench_default.json")))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, TAG + "bench_default.json"))
shutil.copy(os.path.join(SRC, "pmc_mfma_busy.txt"), os.path.join(DST, TAG + "pmc_mfma_busy.txt"))
r = bd["roofline"]
o = r.get("other_large_tile_kernel", {})
out += ["Un-profiled default run: **%.1f imgs/s, %.2f ms/step** (mean of %d bracketed steps; median %.2f ms, p10/p90 %s; consistency "
        "branch skipped in %s steps). Dominant kernel `%s`: %.1f TFLOP/s algorithmic over %d launches/step = **%.3f of %.0f** (the "
        "6-product line of the earlier rounds: %.3f of 417; %.2f x the fp32-input MFMA peak; %.1f %% of step time%s); the other "
        "large-tile kernel `%s`: %.1f TFLOP/s = %.3f over %d launches/step. Same step on the fp32-input MFMA (mode 0): %.1f imgs/s, "
        "%.1f ms/step. CPU baseline (oracle, %d threads): %.3f imgs/s." % (
            bd["value"], bd["ms_per_step"], bd["steps"], bd["median_ms_per_step"], bd["p10_p90_ms_per_step"],
            bd["config"]["consistency_branch_skipped_steps"], r["kernel"].split(" ")[0], r["achieved"], r["launches_per_step"],
            r["frac"], r["peak"], r.get("six_product_line", {}).get("frac", float("nan")), r.get("vs_fp32_mfma_peak", float("nan")),
            100 * r["share_of_step_time"],
            ("; single-stream leg %.1f TFLOP/s = %.3f" % (r["single_stream"]["achieved"], r["single_stream"]["frac"])) if "single_stream" in r else "",
            o.get("kernel", "-").split(" ")[0], o.get("achieved", 0), o.get("frac", 0), o.get("launches_per_step", 0),
            bd.get("fp32_mfma_mode", {}).get("value", float("nan")), bd.get("fp32_mfma_mode", {}).get("ms_per_step", float("nan")),
            bd.get("cpu_baseline", {}).get("cores", 0), bd.get("cpu_baseline", {}).get("value", float("nan"))), ""]
fam, fl = r.get("family"), bd.get("forward_leg")
if fam:
    out += ["**Convolution family** (`roofline.family`: every conv / weight-gradient launch of three single-stream steps, event-bracketed): "
            "bound %.2f ms / measured %.2f ms = **%.3f** over %.0f launches per step, %.2f TFLOP algorithmic per step.  By group: %s." % (
                fam["bound_ms"], fam["time_ms"], fam["frac"], fam["launches_per_step"], fam["algorithmic_tflop_per_step"],
                "; ".join("%s %.2f / %.2f ms (%.2f)" % (k, v["bound_ms"], v["time_ms"], v["frac"]) for k, v in list(fam["groups"].items())[:10])), ""]
if fl:
    out += ["**Forward leg** (`forward_leg`: teacher forward_teacher + student supervised + unsupervised forwards, no autograd, one stream, "
            "12 image-forwards): %.2f ms per pass (device %.2f), %.2f TFLOP algorithmic -> **%.1f TFLOP/s = %.3f of 833** (3 products per "
            "multiply) = %.2f x the fp32-input MFMA peak of 157.3; the leg's own convolution bound %.2f ms of %.2f ms measured (%.3f)." % (
                fl["ms_per_pass"], fl["device_ms_per_pass"], fl["algorithmic_tflop_per_pass"], fl["achieved_tflops"], fl["frac_of_833"],
                fl["frac_of_fp32_mfma_peak_157"], fl["conv_family_of_this_leg"]["bound_ms"], fl["conv_family_of_this_leg"]["time_ms"],
                fl["conv_family_of_this_leg"]["frac"]), ""]
for tag in ("f16x2", "bf16x3", "mode0"):
    path = os.path.join(SRC, "kernel_stats_%s.csv" % tag)
    if not os.path.exists(path):
        continue
    rows = list(csv.DictReader(open(path)))
    for f in ("kernel_stats_%s.csv", "bench_under_rocprof_%s.json", "pmc_FETCH_SIZE_by_kernel_%s.csv", "pmc_WRITE_SIZE_by_kernel_%s.csv"):
        if os.path.exists(os.path.join(SRC, f % tag)):
            shutil.copy(os.path.join(SRC, f % tag), os.path.join(DST, TAG + f % tag))
    b = json.load(open(os.path.join(SRC, "bench_under_rocprof_%s.json" % tag)))
    nsteps = sum(int(x["Calls"]) for x in rows if "ema_kernel" in x["Name"])   # one EMA launch per step: counted, not assumed
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    nl = sum(int(x["Calls"]) for x in rows)
    lib = [x for x in rows if is_library(x["Name"])]
    # the same launches by the profiler's own clock (first wave in to last wave out): the kernel the line names, and for a kernel
    # that finishes split-K shapes in a second launch that launch too (bench.py's brackets hold both)
    kname = b["roofline"]["kernel"].split("<")[0].split(" ")[0]
    pref = KERN[tag].get("fwd4" if "strip" in kname else "fwd1", kname + "<")
    kr = [x for x in rows if pref in x["Name"]]
    kcalls, kns = sum(int(x["Calls"]) for x in kr), sum(float(x["TotalDurationNs"]) for x in kr)
    prof_avg = kns / max(kcalls, 1) / 1e6
    flop = b["roofline"].get("algorithmic_flop_per_launch")
    out += ["## %s -- %s" % (tag, ARITH[tag]), "",
            "bench line under the profiler: %.2f imgs/s, %.1f ms/step; roofline (`%s`): %.1f TFLOP/s (frac %.3f), avg launch %.4f ms "
            "by the event brackets on the launch stream; rocprofv3's own average over the same %d launches: **%.4f ms**%s -- the brackets "
            "start when the stream reaches the launch and so hold its wait for free CUs beside the other two streams, the profiler's "
            "duration starts with the first wave"
            % (b["value"], b["ms_per_step"], b["roofline"]["kernel"].split(" ")[0], b["roofline"]["achieved"], b["roofline"]["frac"],
               b["roofline"]["avg_launch_ms"], kcalls, prof_avg,
               (" (= %.1f TFLOP/s, %.3f of %.0f)" % (flop / prof_avg / 1e9, flop / prof_avg / 1e9 / b["roofline"]["peak"], b["roofline"]["peak"])) if flop and prof_avg else ""), "",
            "%d steps traced (= `ema_kernel` launches): %.1f ms of kernel time = **%.1f ms/step** (sum over all streams), **%d launches/step**, "
            "of which library (ATen / rocprim / runtime copies) %d launches and %.2f ms per step" % (
                nsteps, tot / 1e6, tot / 1e6 / max(nsteps, 1), nl // max(nsteps, 1), sum(int(x["Calls"]) for x in lib) // max(nsteps, 1),
                sum(float(x["TotalDurationNs"]) for x in lib) / 1e6 / max(nsteps, 1)), "",
            "| kernel | calls/step | ms/step | avg us | % |", "|---|---|---|---|---|"]
    for x in rows[:20]:
        out.append("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (short(x["Name"]), int(x["Calls"]) / max(nsteps, 1),
                                                           float(x["TotalDurationNs"]) / 1e6 / max(nsteps, 1),
                                                           float(x["AverageNs"]) / 1e3, float(x["Percentage"])))
    fe_p, wr_p = os.path.join(SRC, "pmc_FETCH_SIZE_by_kernel_%s.csv" % tag), os.path.join(SRC, "pmc_WRITE_SIZE_by_kernel_%s.csv" % tag)
    if os.path.exists(fe_p) and os.path.exists(wr_p):
        fe, wr = list(csv.DictReader(open(fe_p))), list(csv.DictReader(open(wr_p)))
        tm = traffic["by_mode"].setdefault(MODE_OF[tag], {})
        for key, name in KERN[tag].items():
            f1 = [x for x in fe if name in x["kernel"]]
            w1 = [x for x in wr if name in x["kernel"]]
            k1 = [x for x in rows if name in x["Name"]]
            if not (f1 and w1 and k1):
                continue
            nd = sum(int(x["dispatches"]) for x in f1)
            fkb = sum(float(x["FETCH_SIZE_sum"]) for x in f1) / nd
            wkb = sum(float(x["WRITE_SIZE_sum"]) for x in w1) / max(sum(int(x["dispatches"]) for x in w1), 1)
            avg_us = sum(float(x["TotalDurationNs"]) for x in k1) / sum(int(x["Calls"]) for x in k1) / 1e3
            tb = (2.0 * fkb + wkb) * 1024
            tm["traffic_bytes_per_launch_" + key] = tb
            tm["kernel_" + key] = name
            tm["dispatches_" + key] = nd
            live, alg, rr = None, None, b["roofline"]
            if name.split("<")[0] in rr["kernel"]:
                live, alg = rr["avg_launch_ms"] * 1e3, rr["algorithmic_bytes_per_launch"]
            elif "other_large_tile_kernel" in rr and name.split("<")[0] in rr["other_large_tile_kernel"]["kernel"]:
                live = rr["other_large_tile_kernel"]["avg_launch_ms"] * 1e3
            out += ["", "`%s`: rocprof average %.1f us per launch%s. PMC per launch (%s dispatches): FETCH_SIZE %.0f KB (x2 gfx950 correction = "
                    "%.1f MB), WRITE_SIZE %.0f KB -> fabric traffic %.1f MB%s." % (
                        name + ("...>" if name.endswith("<") else ""), avg_us,
                        (" vs %.1f us measured live by bench.py with events on the launch stream, same command (its brackets also hold "
                         "the split-K finish launch of the call)" % live) if live else "",
                        nd, fkb, 2 * fkb * 1024 / 1e6, wkb, tb / 1e6,
                        (" vs %.1f MB algorithmic (input + weights + output + the fused epilogue's operands, each once) = %.2f x" % (alg / 1e6, tb / alg)) if live and alg else "")]
        if "traffic_bytes_per_launch_fwd1" in tm:
            tm["traffic_bytes_per_launch"] = tm["traffic_bytes_per_launch_fwd1"]
    out.append("")
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
extra = [("split_sites.txt", "the launches of one step that still run a plane-split pass in front, with the producer of the tensor they split (round 6: 92 -> 10)"),
         ("whatif.txt", "the step with a family of launches skipped (timing only): what the weight gradients, the optimiser tail and the second stream cost the step"),
         ("conv_table.txt", "every convolution call of a step by shape (single-stream, event-bracketed): time, TFLOP/s, MFMA / HBM bound"),
         ("host_device_phases.txt", "host issue time and device arrival time of every phase of one overlapped step (no profiler)"),
         ("library_op_sites.txt", "which lines of the package still issue library (ATen) operations in a step, by count (both threads)"),
         ("call_hist.txt", "the C-ABI calls of a step by symbol (`library_calls_per_step` of the bench line), direct and replayed from launch plans"),
         ("step_series_bench.txt", "per-step ms over 120 steps with the bench's frozen learning rate"),
         ("clock_under_load.txt", "shader clock and board power of the GPU under the dominant kernel back to back, the fp32-input MFMA kernel, an HBM copy, the bench step"),
         ("precision_f16x2.txt", "time and error against fp64 of the default arithmetic, the 3-term bf16 split and the fp32-input MFMA on the strip and tiled shapes, for activation-like, gradient-like, extreme-scale and outlier-laden operands; launches of a step on the fp16 split and the tensors that still take a reduction pass of their own"),
         ("bench_bf16x3.json", "`python bench.py --bf16x3`: the round-2 default arithmetic (3-term bf16 split, 6 products) on this round's code"),
         ("bench_bf16.json", "`python bench.py --bf16`: bf16 products + bf16 activation storage"),
         ("bench_bf16_irnet.json", "`python bench.py --bf16 --irnet` = BASELINE configs[4] on one GPU"),
         ("bench_irnet.json", "`python bench.py --irnet`: IR-Net on, fp32-grade arithmetic"),
         ("bench_rccl_world1.json", "`MMT_FORCE_DIST=1 MMT_DIST_TRACE=1 torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the RCCL path at world size 1 with the bucketed exchange; `dist_trace` = per-piece issue / arrival times")]
out += ["## How the step is put together; other configurations", ""]
for f, what in extra:
    if os.path.exists(os.path.join(SRC, f)) and os.path.getsize(os.path.join(SRC, f)) > 0:
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, TAG + f))
        line = "* `profiles/%s%s` -- %s" % (TAG, f, what)
        if f.endswith(".json"):
            try:
                j = json.loads(open(os.path.join(SRC, f)).read().strip().splitlines()[-1])
                line += ": **%.1f imgs/s, %.2f ms/step** (median %.2f; skipped-branch steps %s)" % (
                    j["value"], j["ms_per_step"], j["median_ms_per_step"], j["config"].get("consistency_branch_skipped_steps"))
                if "dist_trace" in j:
                    t = j["dist_trace"]
                    line += "; %d of %d pieces (%.0f of %.0f MB) issued before the backward pass ended at %.1f ms" % (
                        t["pieces_sent_before_backward_end"], len(t["pieces"]), sum(q["mbytes"] for q in t["pieces"][:t["pieces_sent_before_backward_end"]]),
                        t["grad_mbytes"], t["backward_end_ms"])
            except Exception as e:
                line += " (unreadable: %s)" % e
        out.append(line)
out.append("")
out += ["## MFMA-busy (single-stream SQ pass)", "", "```"] + [l.rstrip() for l in open(os.path.join(SRC, "pmc_mfma_busy.txt")) if "mfma_busy_fraction" in l or l.startswith("#")] + ["```", ""]
hist = open(os.path.join(DST, "r06_history.md")).read() if os.path.exists(os.path.join(DST, "r06_history.md")) else ""
open(os.path.join(DST, "r06_summary.md"), "w").write("\n".join(out) + "\n" + hist)
print("\n".join(out)[:6000])
