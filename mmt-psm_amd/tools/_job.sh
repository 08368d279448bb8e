cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j44
for tag in on1 off1 on2 off2 on3 off3; do
  case $tag in on*) export MMT_DET_TENSOR=0;; off*) export MMT_DET_TENSOR=1;; esac
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j44/bench_$tag.json 2>gpurun_out/j44/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j44/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['losses']['mt_fg_loss'])" || tail -3 gpurun_out/j44/err_$tag.txt
done
