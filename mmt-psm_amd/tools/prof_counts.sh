cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MMT_BENCH_NO_FP32_LEG=1
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --profile-steps 1 > $R/gpurun_out/bench_under_rocprof.json 2> /tmp/err.txt
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_r02_mid.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/kernel_stats_r02_mid.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); nl = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", nl)
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:45]:
    print("%6s %9.2f ms %8.1f us  %s" % (r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
