cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_data_parallel_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "evaluator or bench or two_rank or rccl" > gpurun_out/t_new.log 2>&1; echo rc=$? >> gpurun_out/t_new.log
