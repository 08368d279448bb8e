cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j53
SO=mmt-psm_amd/maskrcnn_benchmark/_hip.cpython-310-x86_64-linux-gnu.so
python -c "
import sys; sys.path.insert(0,'mmt-psm_amd')
from maskrcnn_benchmark import _hip; print(_hip.__file__)"
python mmt-psm_amd/tools/host_call_cost.py 2>/dev/null | head -9
for tag in cy1 cy2; do
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j53/bench_$tag.json 2>gpurun_out/j53/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j53/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['losses']['mt_fg_loss'])" || tail -3 gpurun_out/j53/err_$tag.txt
done
mv $SO /tmp/hip_so.bak
python mmt-psm_amd/tools/host_call_cost.py 2>/dev/null | head -9
for tag in py1 py2; do
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j53/bench_$tag.json 2>gpurun_out/j53/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j53/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'], d['p10_p90_ms_per_step'], d['config']['losses']['mt_fg_loss'])" || tail -3 gpurun_out/j53/err_$tag.txt
done
mv /tmp/hip_so.bak $SO
MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --profile-steps 0 > gpurun_out/j53/bench_cy3.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/j53/bench_cy3.json'));print('cy3', d['ms_per_step'], d['median_ms_per_step'])"
