# two-stream timeline of the default bench step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MMT_BENCH_NO_FP32_LEG=1
rm -rf /tmp/pt
env ${EXTRA:-X=1} rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 3 --profile-steps 0 > /tmp/b.json 2> /tmp/err.txt
tail -1 /tmp/b.json | cut -c1-160
python3 $R/mmt-psm_amd/tools/stream_timeline.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) ${1:-}
