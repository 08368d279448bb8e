cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j28
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_f16x2_gpu.py tests/test_model_gpu.py tests/test_train_step_gpu.py tests/test_bf16_storage_gpu.py -m gpu -q -x > gpurun_out/j28/pytest.txt 2>&1
tail -4 gpurun_out/j28/pytest.txt
for i in 1 2; do
MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > gpurun_out/j28/bench$i.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/j28/bench$i.json'));print(d['ms_per_step'], d['median_ms_per_step'])"
done
