cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j4
(timeout 1500 python -m pytest tests/test_f16x2_gpu.py tests/test_fullsize_properties.py tests/test_hip_kernels.py -m gpu -q 2>&1 | tail -60) > gpurun_out/j4/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/j4/bench_default.json 2> gpurun_out/j4/bench_default.err
MMT_WGRAD_F16_MIN=2000000000 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/j4/bench_nowgradpass.json 2>/dev/null
timeout 300 python mmt-psm_amd/tools/host_profile.py > gpurun_out/j4/host_profile.txt 2>&1
timeout 300 python mmt-psm_amd/tools/host_phases.py 2>/dev/null | tail -40 > gpurun_out/j4/host_phases.txt
tail -5 gpurun_out/j4/pytest.txt; cut -c1-300 gpurun_out/j4/bench_default.json;  cut -c1-300 gpurun_out/j4/bench_nowgradpass.json
