cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pm() {
  rm -rf /tmp/pq; N=2 CIN=256 HW=256 COUT=256 K=3 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pq -o t -- python mmt-psm_amd/tools/wgrad_dbg.py > /tmp/pq.log 2>&1 || tail -5 /tmp/pq.log
  f=$(find /tmp/pq -name "*counter_collection.csv" | head -1)
  python - "$f" <<'P'
import csv,sys,collections
agg=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_pipe' in r['Kernel_Name']:
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,v) in agg.items(): print("%-32s %14.0f per launch (%d)"%(k, v/n, n))
P
}
pm SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pm SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM
pm SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES
pm SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES
pm SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
