cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j38
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_hip_kernels.py tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -5
run() {
  rm -rf /tmp/gs; N=$1 CIN=$2 COUT=$3 HW=$4 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs -o t -- python mmt-psm_amd/tools/glds_scaling.py > /tmp/gs.log 2>&1
  f=$(find /tmp/gs -name "*kernel_stats.csv" | head -1)
  python - "$f" "$@" <<'P'
import csv,sys
N,CIN,COUT,HW=map(int,sys.argv[2:6])
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls'])>=20 and ('glds' in r['Name'] or 'rows' in r['Name'] or 'finish' in r['Name']):
        us=float(r['AverageNs'])/1e3
        print("N=%2d %4d->%4d @%d  %-60s %.1f us  %.0f TF"%(N,CIN,COUT,HW,r['Name'][28:88],us,2.0*N*HW*HW*CIN*COUT/us/1e6))
P
}
MMT_ROWS=0 run 2 1024 256 64; MMT_ROWS=0 run 4 1024 256 64; MMT_ROWS=0 run 8 1024 256 64; MMT_ROWS=0 run 2 512 256 64; run 2 256 1024 64; run 2 2048 512 32
for tag in a b c; do
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > gpurun_out/j38/bench_$tag.json 2>gpurun_out/j38/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j38/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['config']['losses']['mt_fg_loss'], d['roofline']['other_large_tile_kernel']['achieved'], d['roofline']['achieved'])" || tail -3 gpurun_out/j38/err_$tag.txt
done
