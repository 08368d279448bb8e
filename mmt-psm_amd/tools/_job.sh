cd /tmp && export TMPDIR=/tmp
for parts in 16 4; do
rm -rf /tmp/pt
MMT_FINISH_PARTS=$parts N=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/mmt-psm_amd/tools/small_conv_trace.py > /dev/null 2>&1
echo "PARTS=$parts"
python3 - $(find /tmp/pt -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "finish" in n:
        print("  %-60s calls %4s avg %7.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_f16x2_gpu.py -m gpu -q -x 2>&1 | tail -2
for parts in 16 4 16 4; do
MMT_FINISH_PARTS=$parts MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > /tmp/b.json 2>/dev/null
python -c "
import json
d=json.load(open('/tmp/b.json'));print('parts $parts', d['ms_per_step'], d['median_ms_per_step'])"
done
