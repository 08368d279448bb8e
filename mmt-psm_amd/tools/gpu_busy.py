"""GPU busy fraction from a rocprofv3 kernel trace: union of kernel intervals / wall span of the last steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# the last N kernels (the timed steps; argv[2] = kernels to keep)
iv = iv[-int(sys.argv[2]):]
busy, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = ce - iv[0][0]
print("span %.1f ms, busy %.1f ms (%.1f %%), %d kernels, %d gaps: total %.1f ms, >20us: %d (%.1f ms), >100us: %d (%.1f ms)" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, len(iv), len(gaps), sum(gaps) / 1e6,
    sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6,
    sum(1 for g in gaps if g > 100000), sum(g for g in gaps if g > 100000) / 1e6))
