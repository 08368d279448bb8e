cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_step_gpu.py -x -q -m gpu -k "supervised_step or bench_size" -s > gpurun_out/t_sup.log 2>&1; echo rc=$? >> gpurun_out/t_sup.log
MMT_BENCH_NO_FP32_LEG=1 python bench.py --supervised --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_supervised.json 2> gpurun_out/bench_supervised.err
python mmt-psm_amd/tools/f16_stats.py 2>/dev/null | grep -v amdgpu.ids | tail -16 > gpurun_out/f16_stats.txt
