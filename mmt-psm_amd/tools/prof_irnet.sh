cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MMT_BENCH_NO_FP32_LEG=1
for f in "" "--irnet"; do
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --profile-steps 1 $f > /tmp/b.json 2> /tmp/err.txt
python3 - <<PY
import csv, json, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)[0])))
steps = 6 + 2 + 1 + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows); nl = sum(int(r["Calls"]) for r in rows)
print("== '$f' kernel %.1f ms/step, %d launches/step" % (tot / 1e6 / steps, nl // steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6.0f /step %8.2f ms/step %8.1f us  %s" % (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
done
