#!/bin/bash
# usage: pmc_one.sh "<counters>" <cmd...>   -> per-kernel sums of the counters (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp
ctr="$1"; shift
d=/tmp/pmc_$RANDOM
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- "$@" > /tmp/pmc_out.txt 2>&1
f=$(find $d -name "*counter_collection.csv" | head -1)
python3 - $f <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:48]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:4]:
    print(k)
    for c,x in sorted(v.items()): print("   %-30s %.4g" % (c,x))
PY
