#!/bin/bash
# Regenerates profiles/ evidence on the GPU box (writes to gpurun_out/prof; summarize_profiles.py copies what is judged into
# profiles/r06_*): un-profiled default bench line; per arithmetic (f16x2 = default two-term fp16 split, bf16x3 = MMT_F16X2=0,
# mode0 = fp32-input MFMA) rocprofv3 kernel stats of the same command; two separate PMC passes (FETCH_SIZE / WRITE_SIZE)
# reduced per kernel for the default and mode 0; one SQ pass (MFMA busy); the per-shape conv table; other configurations.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_default.json 2> /tmp/err_default.txt
export MMT_BENCH_NO_FP32_LEG=1 MMT_BENCH_NO_FAMILY_LEG=1   # (the profiled runs hold the step alone)
reduce() {  # csv of a --pmc pass -> per-kernel table of one counter
python3 - "$1" "$2" <<'PY'
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
}
for TAG in f16x2 bf16x3 mode0; do
  unset MMT_CONV_PRECISION MMT_F16X2
  [ $TAG = bf16x3 ] && export MMT_F16X2=0
  [ $TAG = mode0 ] && export MMT_CONV_PRECISION=0
  rm -rf /tmp/ps$TAG
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$TAG -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --profile-steps 5 > $OUT/bench_under_rocprof_$TAG.json 2> /tmp/err_$TAG.txt
  cp $(find /tmp/ps$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$TAG.csv
  if [ $TAG != bf16x3 ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pp
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-steps 1 > /dev/null 2>&1
      reduce $(find /tmp/pp -name "*counter_collection.csv" | head -1) $C > $OUT/pmc_${C}_by_kernel_$TAG.csv
    done
  fi
done
unset MMT_CONV_PRECISION MMT_F16X2
rm -rf /tmp/pq
MMT_OVERLAP_TEACHER=0 MMT_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/pq -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-steps 1 > /dev/null 2>&1
python3 - $(find /tmp/pq -name "*counter_collection.csv" | head -1) > $OUT/pmc_mfma_busy.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA (one pass, --kernel-trace only) of")
print("# MMT_OVERLAP_TEACHER=0 MMT_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 2 --warmup 1; sums over all dispatches of a kernel;")
print("# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:10]
for k, c in rows:
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4 * c.get("SQ_BUSY_CU_CYCLES", 1), 1)
    print("%s   mfma_busy_fraction %.3f" % (k[:70], busy))
    for n in sorted(c):
        print("   %-28s %.4g" % (n, c[n]))
PY
# how the step is put together (no profiler unless noted), other configurations
python $R/mmt-psm_amd/tools/split_sites.py 2>/dev/null | grep -v amdgpu.ids | cut -c1-260 > $OUT/split_sites.txt
python $R/mmt-psm_amd/tools/whatif.py 2>/dev/null | grep -v amdgpu.ids > $OUT/whatif.txt
MMT_WGRAD_STREAM=0 python $R/mmt-psm_amd/tools/conv_table.py > $OUT/conv_table.txt 2>/dev/null
python $R/mmt-psm_amd/tools/host_phases.py 2>/dev/null | tail -36 > $OUT/host_device_phases.txt
python $R/mmt-psm_amd/tools/op_sites.py 2>/dev/null | grep -v amdgpu.ids | head -60 > $OUT/library_op_sites.txt
python $R/mmt-psm_amd/tools/call_hist.py 5 2>/dev/null | grep -v amdgpu.ids > $OUT/call_hist.txt
STEPS=120 python $R/mmt-psm_amd/tools/step_series.py 2>/dev/null | tail -3 > $OUT/step_series_bench.txt
python $R/bench.py --bf16x3 --no-cpu-baseline > $OUT/bench_bf16x3.json 2>/dev/null
python $R/bench.py --bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>/dev/null
python $R/bench.py --bf16 --irnet --no-cpu-baseline > $OUT/bench_bf16_irnet.json 2>/dev/null
python $R/bench.py --irnet --no-cpu-baseline > $OUT/bench_irnet.json 2>/dev/null
MMT_FORCE_DIST=1 MMT_DIST_TRACE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29751 $R/bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_rccl_world1.json
python $R/mmt-psm_amd/tools/clock_under_load.py 2>/dev/null | grep -v amdgpu.ids > $OUT/clock_under_load.txt
python $R/mmt-psm_amd/tools/bench_f16x2.py 2>/dev/null | grep -v amdgpu.ids > $OUT/precision_f16x2.txt
python $R/mmt-psm_amd/tools/bench_f16x2_glds.py 2>/dev/null | grep -v amdgpu.ids >> $OUT/precision_f16x2.txt
python $R/mmt-psm_amd/tools/f16_stats.py 2>/dev/null | grep -v amdgpu.ids >> $OUT/precision_f16x2.txt
ls -la $OUT
