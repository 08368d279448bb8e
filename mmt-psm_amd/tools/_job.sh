cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fullsize_properties.py -x -q -m gpu -k bucketed > gpurun_out/t_dp.log 2>&1; echo rc=$? >> gpurun_out/t_dp.log
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" > gpurun_out/ab_rccl.txt 2>&1
export MMT_BENCH_NO_FP32_LEG=1
for pr in 0 -1 1; do for i in 1 2; do
MMT_WGRAD_PRIO_EXPERIMENT=$pr MMT_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2975$i bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rccl1 wgrad prio $pr', d['ms_per_step'], d['median_ms_per_step'])"
done; done >> gpurun_out/ab_rccl.txt 2>&1
for pr in -1 1; do MMT_WGRAD_PRIO_EXPERIMENT=$pr python bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nodist wgrad prio $pr', d['ms_per_step'], d['median_ms_per_step'])"; done >> gpurun_out/ab_rccl.txt 2>&1
