cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j19
timeout 900 python -m pytest tests/test_train_step_gpu.py -m gpu -q -x > gpurun_out/j19/pytest.txt 2>&1
tail -3 gpurun_out/j19/pytest.txt
for v in "default" "irnet --irnet" "irnet_ns --irnet"; do
  set -- $v; tag=$1; shift
  if [ $tag = irnet_ns ]; then export MMT_WGRAD_STREAM=0; else unset MMT_WGRAD_STREAM; fi
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 "$@" > gpurun_out/j19/bench_$tag.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/j19/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'])"
done
unset MMT_WGRAD_STREAM
cd /tmp && export TMPDIR=/tmp
export MMT_BENCH_NO_FP32_LEG=1
rm -rf /tmp/ps
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $GRAFT_REPO_ROOT/bench.py --irnet --no-cpu-baseline --steps 5 --warmup 2 --profile-steps 5 > $GRAFT_REPO_ROOT/gpurun_out/j19/bench_under_rocprof_irnet.json 2> /tmp/err.txt
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/j19/kernel_stats_irnet.csv
