cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j30
timeout 900 python -m pytest tests/test_train_step_gpu.py -m gpu -q -x -k "pair_forward" > gpurun_out/j30/pytest.txt 2>&1
tail -15 gpurun_out/j30/pytest.txt
