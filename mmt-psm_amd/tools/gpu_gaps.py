"""largest idle gaps of the GPU in a rocprofv3 kernel trace (last argv[2] kernels), with the kernels around them"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
gaps = []
ce, last = int(rows[0]["End_Timestamp"]), rows[0]
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > ce:
        gaps.append((s - ce, last["Kernel_Name"][:50], r["Kernel_Name"][:50]))
    if e > ce:
        ce, last = e, r
by = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g > 15000:
        by[(a, b)][0] += 1
        by[(a, b)][1] += g
span = ce - int(rows[0]["Start_Timestamp"])
print("span %.1f ms; gaps > 15 us: %d, %.1f ms" % (span / 1e6, sum(v[0] for v in by.values()), sum(v[1] for v in by.values()) / 1e6))
for (a, b), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%3d x  %7.1f us avg   after %-50s before %s" % (n, t / n / 1e3, a, b))
