cd /root/repo
python mmt-psm_amd/tools/rows_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/rows_ab4.txt
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1
tail -3 gpurun_out/gpu_tests.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'])"
done > gpurun_out/rows_ab_step4.txt
cat gpurun_out/rows_ab4.txt gpurun_out/rows_ab_step4.txt
