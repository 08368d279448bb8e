# A/B of the tap-strip kernel's tile form (MMT_STRIP_TW64): fabric read traffic per launch (FETCH_SIZE, own pass) -- timing is in bench.py's roofline legs
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/split
python $R/mmt-psm_amd/tools/call_hist.py 5 > $R/gpurun_out/split/call_hist.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export MMT_BENCH_NO_FP32_LEG=1 MMT_BENCH_NO_FAMILY_LEG=1
for arm in base tw64; do
  case $arm in tw64*) export MMT_STRIP_TW64=1;; *) unset MMT_STRIP_TW64;; esac
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
done 2>&1 | tee $R/gpurun_out/split/tw64_traffic.txt
head -3 $R/gpurun_out/split/call_hist.txt
