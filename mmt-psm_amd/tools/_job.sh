cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gputest_d.log 2>&1; echo rc=$? >> gpurun_out/gputest_d.log
