cd /root/repo
run() { python mmt-psm_amd/tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
MMT_KG=0 run S3_kg0
MMT_KG=1 run S3_kg1
MMT_KG=0 MMT_LIB=mmt-psm_amd/libS4.so run S4_kg0
MMT_KG=0 MMT_LIB=mmt-psm_amd/libS5.so run S5_kg0
done > gpurun_out/stages_step.txt
cat gpurun_out/stages_step.txt
