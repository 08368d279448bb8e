cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j42
for tag in on1 off1 on2 off2 on3 off3 on4 off4; do
  case $tag in on*) export MMT_FORK_SUM=1;; off*) export MMT_FORK_SUM=0;; esac
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --profile-steps 0 > gpurun_out/j42/bench_$tag.json 2>gpurun_out/j42/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j42/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'])" || tail -3 gpurun_out/j42/err_$tag.txt
done
