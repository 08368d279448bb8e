#!/bin/bash
# Regenerates profiles/ evidence on the GPU box (writes to gpurun_out/prof; copy what is wanted into profiles/):
#   rocprofv3 kernel stats of `bench.py`, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) reduced per kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MMT_BENCH_NO_FP32_LEG=1
for MODE in 3 0; do
  export MMT_CONV_PRECISION=$MODE
  rm -rf /tmp/ps$MODE
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$MODE -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_under_rocprof_mode$MODE.json 2> /tmp/err_$MODE.txt
  cp $(find /tmp/ps$MODE -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_mode$MODE.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pp
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
    python3 - $(find /tmp/pp -name "*counter_collection.csv" | head -1) $C > $OUT/pmc_${C}_by_kernel_mode$MODE.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print("kernel,dispatches,%s_sum,%s_per_dispatch" % (sys.argv[2], sys.argv[2]))
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('"%s",%d,%.1f,%.1f' % (k[:120], n, v, v / n))
PY
  done
done
ls -la $OUT
