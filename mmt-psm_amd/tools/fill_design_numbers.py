"""DESIGN.md section 5 (round 6) / section 6 quote numbers of profiles/r06_*; after a new profile run this replaces the ones written
last time (kept in profiles/r06_design_numbers.json) by the new ones, inside those two passages only."""
import csv, json, os
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
P = lambda f: os.path.join(ROOT, "profiles", f)
old = json.load(open(P("r06_design_numbers.json")))
d = json.load(open(P("r06_bench_default.json")))
r = json.load(open(P("r06_bench_rccl_world1.json")))
rows = list(csv.DictReader(open(P("r06_kernel_stats_f16x2.csv"))))
nsteps = sum(int(x["Calls"]) for x in rows if "ema_kernel" in x["Name"])
kms = sum(float(x["TotalDurationNs"]) for x in rows) / 1e6 / nsteps
nl = sum(int(x["Calls"]) for x in rows) // nsteps
f, fam = d["forward_leg"], d["roofline"]["family"]
rep = {"@R6_VALUE@": "%.1f" % d["value"], "@R6_MS@": "%.2f" % d["ms_per_step"], "@R6_MED@": "%.2f" % d["median_ms_per_step"],
       "@R6_KMS@": "%.1f" % kms, "@R6_LAUNCHES@": "%d" % nl, "@R6_FAM@": "%.3f" % fam["frac"], "@R6_FAMB@": "%.2f" % fam["bound_ms"],
       "@R6_FAMT@": "%.2f" % fam["time_ms"], "@R6_FWDMS@": "%.2f" % f["ms_per_pass"], "@R6_FWDTF@": "%.2f" % f["algorithmic_tflop_per_pass"],
       "@R6_FWDACH@": "%.1f" % f["achieved_tflops"], "@R6_FWD833@": "%.3f" % f["frac_of_833"],
       "@R6_FWD157@": "%.2f" % f["frac_of_fp32_mfma_peak_157"], "@R6_FWDFAM@": "%.2f" % f["conv_family_of_this_leg"]["frac"],
       "@R6_FRAC@": "%.3f (single-stream leg %.3f)" % (d["roofline"]["frac"], d["roofline"]["single_stream"]["frac"]),
       "@R6_RING@": "%.2f" % r["dist_trace"]["predicted_8gpu"]["ring"]["exposed_comm_ms"],
       "@R6_DIRECT@": "%.2f" % r["dist_trace"]["predicted_8gpu"]["direct"]["exposed_comm_ms"]}
s = open(os.path.join(ROOT, "DESIGN.md")).read()
a, b = s.index("### Round 6 (1 × MI355X"), s.index("### Round 5 (1 × MI355X")
reg = s[a:b]
for k in sorted(old, key=lambda k: -len(old[k])):
    if k in ("@R6_RING@", "@R6_DIRECT@"):
        continue
    assert old[k] in reg, (k, old[k])
    reg = reg.replace(old[k], "\x00" + k + "\x00")
for k in rep:
    reg = reg.replace("\x00" + k + "\x00", rep[k])
s = s[:a] + reg + s[b:]
a, b = s.index("Round 6 (VERDICT r5 item 9)"), s.index("## 7. Status of SURVEY")
s = s[:a] + s[a:b].replace("≈ %s ms (ring) / %s ms (direct)" % (old["@R6_RING@"], old["@R6_DIRECT@"]),
                            "≈ %s ms (ring) / %s ms (direct)" % (rep["@R6_RING@"], rep["@R6_DIRECT@"])) + s[b:]
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
json.dump(rep, open(P("r06_design_numbers.json"), "w"), indent=1)
print(rep)
