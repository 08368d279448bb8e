cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j51
for tag in d1 f1 d2 f2 d3 f3; do
  case $tag in d*) unset MMT_TEACHER_FIRST;; f*) export MMT_TEACHER_FIRST=1;; esac
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j51/bench_$tag.json 2>gpurun_out/j51/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j51/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'])" || tail -3 gpurun_out/j51/err_$tag.txt
done
