cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j12
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/j12/pytest.txt
tail -8 gpurun_out/j12/pytest.txt
