cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j13
timeout 1800 python -m pytest tests/test_config4_gpu.py tests/test_irnet_gpu.py tests/test_bf16_storage_gpu.py -m gpu -q > gpurun_out/j13/pytest.txt 2>&1
tail -3 gpurun_out/j13/pytest.txt
