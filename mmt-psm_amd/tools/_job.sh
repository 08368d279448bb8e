cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j57
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python mmt-psm_amd/tools/f16_stats.py 2>/dev/null | grep -A8 "reduction passes" | head -12
for tag in a b; do
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j57/bench_$tag.json 2>gpurun_out/j57/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j57/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['losses']['mt_fg_loss'])" || tail -3 gpurun_out/j57/err_$tag.txt
done
