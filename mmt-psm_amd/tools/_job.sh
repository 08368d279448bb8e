cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j17
timeout 900 python -m pytest tests/test_select_gpu.py tests/test_proposals_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/j17/pytest.txt 2>&1
tail -3 gpurun_out/j17/pytest.txt
for v in "default" "bf16 --bf16" "bf16_irnet --bf16 --irnet" "irnet --irnet"; do
  set -- $v; tag=$1; shift
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 "$@" > gpurun_out/j17/bench_$tag.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/j17/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'])"
done
