"""Per-step picture of the two streams from a rocprofv3 kernel trace (argv[1] = *_kernel_trace.csv): steps are cut at the
EMA kernel; per queue/stream busy time, union busy, both-busy time, and a 1-ms timeline (fraction of each ms each stream
is busy, and the kernel that took most of it)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
key = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 1 else "Queue_Id"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [int(r["End_Timestamp"]) for r in rows if "ema_kernel" in r["Kernel_Name"]]
print("columns:", key, "steps:", len(cuts) - 1)
def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None: cs, ce = s, e
        elif s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(cuts) - 3
for st in range(max(0, len(cuts) - 4), len(cuts) - 1):
    a, b = cuts[st], cuts[st + 1]
    rs = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
    per = collections.defaultdict(list)
    for r in rs: per[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    u = union([iv for l in per.values() for iv in l])
    pu = {k: union(l) for k, l in per.items()}
    print("step %d: wall %.2f ms, union busy %.2f ms (%.1f %%), kernel sum %.2f ms, launches %d; per %s busy: %s; both busy %.2f ms" % (
        st, (b - a) / 1e6, u / 1e6, 100.0 * u / (b - a), sum(e - s for l in per.values() for s, e in l) / 1e6, len(rs), key,
        ", ".join("%s: %.2f ms (%d)" % (k, v / 1e6, len(per[k])) for k, v in sorted(pu.items(), key=lambda kv: -kv[1])),
        (sum(pu.values()) - u) / 1e6))
    if st != which: continue
    ks = sorted(pu, key=lambda k: -pu[k])[:2]
    nb = (b - a) // 1000000 + 1
    for ms in range(nb):
        lo, hi = a + ms * 1000000, a + (ms + 1) * 1000000
        line = "%3d ms |" % ms
        for k in ks:
            occ, names = 0, collections.Counter()
            for r in rs:
                if r[key] != k: continue
                s, e = max(int(r["Start_Timestamp"]), lo), min(int(r["End_Timestamp"]), hi)
                if e > s:
                    occ += e - s
                    names[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:34]] += e - s
            n = sum(1 for r in rs if r[key] == k and lo <= int(r["Start_Timestamp"]) < hi)
            line += " %3d%% %3d launches %-34s |" % (occ // 10000, n, names.most_common(1)[0][0] if names else "-")
        print(line)
