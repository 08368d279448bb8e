cd $GRAFT_REPO_ROOT
export MMT_BENCH_NO_FP32_LEG=1
python - <<'PY' > gpurun_out/ab_fork.txt 2>&1
import json, subprocess, sys, os
for i in range(2):
    for tag, env in (("chain       ", {"MMT_FORK_TAIL_OFF": "1"}), ("fork prio 0 ", {"MMT_FORK_PRIO": "0"}), ("emb only -1 ", {"MMT_FORK_WHICH": "emb"}),
                     ("mask only -1", {"MMT_FORK_WHICH": "mask"}), ("emb only 0  ", {"MMT_FORK_WHICH": "emb", "MMT_FORK_PRIO": "0"}),
                     ("mask only 0 ", {"MMT_FORK_WHICH": "mask", "MMT_FORK_PRIO": "0"}), ("fork prio 1 ", {"MMT_FORK_PRIO": "1"})):
        p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--profile-steps", "1"],
                           env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
        print(tag, d["ms_per_step"], d["median_ms_per_step"], d["p10_p90_ms_per_step"], flush=True)
PY
