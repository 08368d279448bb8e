# A/B of the tap-strip kernel's K order (MMT_STRIP_KORDER=0: kh-major as in rounds 2-5; default slab-major): step / kernel time from
# bench.py's legs, fabric traffic per launch from own FETCH_SIZE / WRITE_SIZE passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/split
mkdir -p $O
cd $R
python -m pytest tests/test_f16x2_gpu.py tests/test_hip_kernels.py tests/test_rb_epilogue_gpu.py -x -q 2>&1 | tail -3
for arm in kh slab kh2 slab2; do
  case $arm in kh*) export MMT_STRIP_KORDER=0;; *) unset MMT_STRIP_KORDER;; esac
  python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/ko_$arm.json 2> $O/ko_$arm.err
done
python - <<'PY'
import json
for arm in ("kh","slab","kh2","slab2"):
    d=json.loads(open("gpurun_out/split/ko_%s.json"%arm).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(arm, "ms/step %.2f median %.2f | strip: launches %s avg %.4f ms frac %.3f single %.3f | family %.3f time %.2f | fwd leg %.2f" % (d["ms_per_step"], d["median_ms_per_step"], r["launches_per_step"], r["avg_launch_ms"], r["frac"], r["single_stream"]["frac"], r["family"]["frac"], r["family"]["time_ms"], d["forward_leg"]["ms_per_pass"]))
PY
cd /tmp && export TMPDIR=/tmp
export MMT_BENCH_NO_FP32_LEG=1 MMT_BENCH_NO_FAMILY_LEG=1
for arm in kh slab; do
  case $arm in kh*) export MMT_STRIP_KORDER=0;; *) unset MMT_STRIP_KORDER;; esac
  for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-steps 1 > /dev/null 2>&1
  python3 - $(find /tmp/pp -name "*counter_collection.csv" | head -1) $arm $C <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[3]:
        continue
    k = r["Kernel_Name"]
    if "strip" not in k: continue
    a = agg[k[:60]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    print(sys.argv[2], sys.argv[3], k, n, "per dispatch %.1f (counter units)" % (v / n))
PY
  done
done 2>&1 | tee $O/korder_traffic.txt
