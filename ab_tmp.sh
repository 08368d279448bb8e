timeout 600 python mmt-psm_amd/tools/bench_pg.py --split 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_pgemm_gpu.py tests/test_f16x2_gpu.py -q -x 2>&1 | tail -3
run() { echo "$1"; env $1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"])"; }
for i in 1 2 3; do
run "MMT_PG_RB=1"
run "MMT_PG_RB=0"
done
