cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j16
timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/j16/pytest.txt 2>&1
tail -5 gpurun_out/j16/pytest.txt
for i in 1 2; do
MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > gpurun_out/j16/bench$i.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/j16/bench$i.json'));print(d['ms_per_step'], d['median_ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
export MMT_BENCH_NO_FP32_LEG=1
rm -rf /tmp/ps
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --profile-steps 5 > $GRAFT_REPO_ROOT/gpurun_out/j16/bench_under_rocprof.json 2> /tmp/err.txt
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/j16/kernel_stats.csv
