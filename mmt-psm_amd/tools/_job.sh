cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_a.log 2>&1; echo rc=$? >> gpurun_out/gputest_a.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1
