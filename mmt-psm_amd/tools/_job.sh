cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j48
timeout 900 python -m pytest tests/test_train_step_gpu.py tests/test_data_parallel_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
for tag in b4a b1a b8a b4b b1b b8b b16; do
  case $tag in b4*) export MMT_WGRAD_BATCH=4;; b1*) export MMT_WGRAD_BATCH=1;; b8*) export MMT_WGRAD_BATCH=8;; b16*) export MMT_WGRAD_BATCH=16;; esac
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j48/bench_$tag.json 2>gpurun_out/j48/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j48/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['losses']['mt_fg_loss'])" || tail -3 gpurun_out/j48/err_$tag.txt
done
