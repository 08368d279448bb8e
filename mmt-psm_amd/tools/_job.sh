cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j23
for v in "default" "irnet --irnet" "nms --irnet" "mask --irnet"; do
  set -- $v; tag=$1; shift
  unset MMT_IRNET_PARTS
  [ $tag = nms ] && export MMT_IRNET_PARTS=nms
  [ $tag = mask ] && export MMT_IRNET_PARTS=mask
  MMT_BENCH_NO_FP32_LEG=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --profile-steps 2 "$@" > gpurun_out/j23/bench_$tag.json 2>gpurun_out/j23/err_$tag.txt
  python -c "
import json
d=json.load(open('gpurun_out/j23/bench_$tag.json'));print('$tag', d['ms_per_step'], d['median_ms_per_step'])" || tail -3 gpurun_out/j23/err_$tag.txt
done
