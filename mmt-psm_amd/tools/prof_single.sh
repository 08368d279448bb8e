# kernel-time sum vs wall time of the step, single-stream (MMT_OVERLAP_TEACHER=0) and overlapped
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MMT_BENCH_NO_FP32_LEG=1
for ov in 0 1; do
rm -rf /tmp/ps
MMT_OVERLAP_TEACHER=$ov rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --profile-steps 1 > /tmp/b.json 2> /tmp/err.txt
python3 - <<PY
import csv, json, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); nl = sum(int(r["Calls"]) for r in rows)
d = json.load(open("/tmp/b.json"))
steps = 10 + 3 + 1 + (0 if $ov == 0 else 1 + 2)   # timed + warm-up + profile leg (+ single-stream leg when overlapped)
conv = sum(float(r["TotalDurationNs"]) for r in rows if "conv" in r["Name"] or "wgrad" in r["Name"] or "pack" in r["Name"] or "split_planes" in r["Name"])
print("overlap=$ov step %.2f ms (under rocprof) | kernel time %.2f ms/step in %d launches/step | conv-family %.2f ms/step | others %.2f" % (
    d["ms_per_step"], tot / 1e6 / steps, nl // steps, conv / 1e6 / steps, (tot - conv) / 1e6 / steps))
PY
done
