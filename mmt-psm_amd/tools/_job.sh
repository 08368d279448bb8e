cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_irnet_gpu.py tests/test_config4_gpu.py -x -q -m gpu 2>&1 | tail -3
