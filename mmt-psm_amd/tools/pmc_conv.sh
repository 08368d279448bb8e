cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 1 0; do
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    rm -rf /tmp/pm; MMT_STRIP=$mode rocprofv3 --pmc $set --kernel-trace -d /tmp/pm -o pm --output-format csv -- python $R/mmt-psm_amd/tools/one_conv.py > /dev/null 2>&1
    python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:40]
        if 'conv' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    print("MMT_STRIP=$mode", k, {c: round(v / cnt[(k, c)] / 1e6, 2) for c, v in acc[k].items()}, "M per launch")
PY
  done
done
