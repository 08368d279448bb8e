cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j15
timeout 900 python -m pytest tests/test_f16x2_gpu.py tests/test_hip_kernels.py -m gpu -q -x > gpurun_out/j15/pytest.txt 2>&1
tail -15 gpurun_out/j15/pytest.txt
MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > gpurun_out/j15/bench.json 2>/dev/null
MMT_ROWS_MIN16=100000 MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 > gpurun_out/j15/bench_norows16.json 2>/dev/null
python -c "
import json
for f in ('bench','bench_norows16'):
    d=json.load(open('gpurun_out/j15/%s.json'%f));print(f, d['ms_per_step'], d['median_ms_per_step'])"
MMT_WGRAD_STREAM=0 timeout 900 python mmt-psm_amd/tools/conv_table.py 2>/dev/null > gpurun_out/j15/conv_table.txt
head -24 gpurun_out/j15/conv_table.txt | cut -c1-150
