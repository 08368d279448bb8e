cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_f16x2_gpu.py tests/test_train_step_gpu.py tests/test_data_parallel_gpu.py -x -q -m gpu > gpurun_out/t_pair.log 2>&1; echo rc=$? >> gpurun_out/t_pair.log
export MMT_BENCH_NO_FP32_LEG=1
for i in 1 2 3; do
MMT_WGRAD_PAIR=0 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('unpaired', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['f16_split_launches'])"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('paired  ', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['f16_split_launches'])"
done > gpurun_out/ab_pair.txt 2>&1
