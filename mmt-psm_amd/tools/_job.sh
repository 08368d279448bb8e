cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_b.log 2>&1; echo rc=$? >> gpurun_out/gputest_b.log
export MMT_BENCH_NO_FP32_LEG=1
for i in 1 2; do
MMT_LIB=mmt-psm_amd/libmmtpsm_base.so python mmt-psm_amd/tools/bench_with_lib.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base ', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'])"
python mmt-psm_amd/tools/bench_with_lib.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('guard', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'])"
done > gpurun_out/ab_guard_step.txt 2>&1
