"""gpurun_out/prof/* (make_profiles.sh) -> profiles/r01_* + profiles/pmc_traffic.json + profiles/r01_summary.md"""
import csv, json, os, shutil, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
DOM = {3: "conv_fwd_glds_kernel<128, 128, 4, 1, 3, 3>", 0: "conv_fwd_kernel<128, 128, 2, 2>"}


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "at::").split("(")[0][:70]


out = ["# Round 1 -- profiles of `python bench.py` on 1 x MI355X (final state of the round)", "",
       "Produced by `mmt-psm_amd/tools/make_profiles.sh` (GPU box) + `mmt-psm_amd/tools/summarize_profiles.py`. Two arithmetic "
       "modes: **mode 3** = default (3-term bf16 split on the bf16 matrix pipe, fp32-grade), **mode 0** = fp32-input MFMA "
       "(`MMT_CONV_PRECISION=0`). Files per mode: `r01_kernel_stats_modeM.csv` (rocprofv3 --kernel-trace --stats of "
       "`bench.py --steps 5 --warmup 2 --no-cpu-baseline`), `r01_bench_under_rocprof_modeM.json`, "
       "`r01_pmc_{FETCH,WRITE}_SIZE_by_kernel_modeM.csv` (two separate --pmc passes, `--steps 2 --warmup 1`); "
       "`r01_bench_default.json` = un-profiled default run; `r01_precision.txt` = error vs fp64 and speed per mode; "
       "`pmc_traffic.json` = what bench.py reports as roofline.traffic.", ""]
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of `python bench.py "
                     "--steps 2 --warmup 1 --no-cpu-baseline`; per-kernel tables profiles/r01_pmc_*_by_kernel_mode*.csv",
           "fetch_correction": 2.0,
           "note": "FETCH_SIZE on gfx950 reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): "
                   "doubled. WRITE_SIZE uncalibrated, taken as is. Infinity-Cache hits are counted, so this is fabric traffic, an "
                   "upper bound on HBM traffic.", "by_mode": {}}
bd = json.load(open(os.path.join(SRC, "bench_default.json")))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, "r01_bench_default.json"))
out += ["Un-profiled default run: **%.1f imgs/s, %.1f ms/step**; dominant kernel %.1f TFLOP/s algorithmic (%.0f executed bf16-MFMA "
        "TFLOP/s, frac %.3f of 2500/6; %.2f x the fp32-MFMA peak); same step in mode 0: %.1f imgs/s, %.1f ms/step, dominant "
        "kernel %.1f TFLOP/s (frac %.3f of 157.3)." % (
            bd["value"], bd["ms_per_step"], bd["roofline"]["achieved"], bd["roofline"].get("executed_mfma_tflops", 0),
            bd["roofline"]["frac"], bd["roofline"].get("vs_fp32_mfma_peak", 0), bd["fp32_mfma_mode"]["value"],
            bd["fp32_mfma_mode"]["ms_per_step"], bd["fp32_mfma_mode"]["dominant_kernel_tflops"],
            bd["fp32_mfma_mode"]["frac_of_fp32_mfma_peak"]), ""]
for mode in (3, 0):
    rows = list(csv.DictReader(open(os.path.join(SRC, "kernel_stats_mode%d.csv" % mode))))
    for f in ("kernel_stats_mode%d.csv", "bench_under_rocprof_mode%d.json", "pmc_FETCH_SIZE_by_kernel_mode%d.csv",
              "pmc_WRITE_SIZE_by_kernel_mode%d.csv"):
        shutil.copy(os.path.join(SRC, f % mode), os.path.join(DST, "r01_" + f % mode))
    b = json.load(open(os.path.join(SRC, "bench_under_rocprof_mode%d.json" % mode)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    nl = sum(int(r["Calls"]) for r in rows)
    out += ["## mode %d" % mode, "",
            "bench line under the profiler: %.2f imgs/s, %.1f ms/step; roofline.achieved %.1f TFLOP/s (frac %.3f), avg launch %.4f ms"
            % (b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"]), "",
            "7 steps traced: %.1f ms of kernel time = %.1f ms/step, %d launches/step" % (tot / 1e6, tot / 7e6, nl // 7), "",
            "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:16]:
        out.append("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    dom = [r for r in rows if DOM[mode] in r["Name"]][0]
    fe = [r for r in csv.DictReader(open(os.path.join(SRC, "pmc_FETCH_SIZE_by_kernel_mode%d.csv" % mode))) if DOM[mode] in r["kernel"]][0]
    wr = [r for r in csv.DictReader(open(os.path.join(SRC, "pmc_WRITE_SIZE_by_kernel_mode%d.csv" % mode))) if DOM[mode] in r["kernel"]][0]
    fkb, wkb = float(fe["FETCH_SIZE_per_dispatch"]), float(wr["WRITE_SIZE_per_dispatch"])
    tb = (2.0 * fkb + wkb) * 1024
    traffic["by_mode"][str(mode)] = {"kernel": DOM[mode], "dispatches": int(fe["dispatches"]), "fetch_size_kb_per_launch": fkb,
                                     "write_size_kb_per_launch": wkb, "traffic_bytes_per_launch": tb}
    fin = [r for r in rows if "conv_splitk_finish_kernel" in r["Name"]]
    fin_note = ""
    if fin:
        fin_note = (" (the event brackets of its %s split-K launches also contain their `conv_splitk_finish_kernel`, %.1f us each: "
                    "+%.1f us on the average)" % (fin[0]["Calls"], float(fin[0]["AverageNs"]) / 1e3,
                                                  float(fin[0]["TotalDurationNs"]) / 1e3 / int(dom["Calls"])))
    out += ["", "Dominant kernel `%s`: rocprof average %.1f us per launch vs %.1f us measured live by bench.py with events on the "
            "launch stream (same command)" + fin_note + ". PMC per launch (%s dispatches): FETCH_SIZE %.0f KB (x2 gfx950 correction = %.1f MB), "
            "WRITE_SIZE %.0f KB -> traffic %.1f MB vs %.1f MB algorithmic (input + weights + output once)."]
    out[-1] = out[-1] % (
                DOM[mode], float(dom["AverageNs"]) / 1e3, b["roofline"]["avg_launch_ms"] * 1e3, fe["dispatches"], fkb,
                2 * fkb * 1024 / 1e6, wkb, tb / 1e6, b["roofline"]["algorithmic_bytes_per_launch"] / 1e6)
    out.append("")
traffic["traffic_bytes_per_launch"] = traffic["by_mode"]["0"]["traffic_bytes_per_launch"]
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
hist = open(os.path.join(DST, "r01_history.md")).read() if os.path.exists(os.path.join(DST, "r01_history.md")) else ""
open(os.path.join(DST, "r01_summary.md"), "w").write("\n".join(out) + "\n" + hist)
print("\n".join(out)[:3000])
