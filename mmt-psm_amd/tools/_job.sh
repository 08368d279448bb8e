cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_c.log 2>&1; echo rc=$? >> gpurun_out/gputest_c.log
python mmt-psm_amd/tools/f16_stats.py 2>/dev/null | grep -v amdgpu.ids | tail -30 > gpurun_out/f16_stats.txt
export MMT_BENCH_NO_FP32_LEG=1
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'])"; done > gpurun_out/bench_c.txt
