"""kernel-time difference between two steps of a rocprofv3 kernel trace (argv: trace.csv stepA stepB; steps cut at ema_kernel)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [int(r["End_Timestamp"]) for r in rows if "ema_kernel" in r["Kernel_Name"]]
def step(i):
    a, b = cuts[i - 1], cuts[i]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s = int(r["Start_Timestamp"])
        if a <= s < b:
            k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
            agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - s
    return agg, (b - a) / 1e6
print("steps:", len(cuts), "walls:", " ".join("%d:%.0f" % (i, (cuts[i] - cuts[i - 1]) / 1e6) for i in range(1, len(cuts))))
A, B = int(sys.argv[2]), int(sys.argv[3])
ga, wa = step(A); gb, wb = step(B)
print("step %d: wall %.2f ms, kernels %.2f ms, %d launches | step %d: wall %.2f ms, kernels %.2f ms, %d launches" % (
    A, wa, sum(v[1] for v in ga.values()) / 1e6, sum(v[0] for v in ga.values()), B, wb, sum(v[1] for v in gb.values()) / 1e6, sum(v[0] for v in gb.values())))
keys = sorted(set(ga) | set(gb), key=lambda k: -abs(gb.get(k, [0, 0])[1] - ga.get(k, [0, 0])[1]))
for k in keys[:25]:
    a, b = ga.get(k, [0, 0]), gb.get(k, [0, 0])
    print("%-70s  %4d x %8.3f ms | %4d x %8.3f ms | %+8.3f" % (k, a[0], a[1] / 1e6, b[0], b[1] / 1e6, (b[1] - a[1]) / 1e6))
