#!/bin/bash
# counters of the C5 rank sweep for two library variants (pop_msym_kernel only)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out
SQ1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ3="SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
for v in "$@"; do
  export DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$v.so
  i=0
  for set in "$SQ1" "$SQ2" "$SQ3"; do i=$((i+1))
    rm -rf $O/x25_${v}_$i
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/x25_${v}_$i -o s -- python3 $R/scratch/c5_bench.py --pop-only --reps 1 > /dev/null 2>&1
  done
done
cd $R
python3 - "$@" <<'PY'
import csv, sys, glob, collections
for v in sys.argv[1:]:
    tot = collections.Counter()
    for i in (1, 2, 3):
        for f in glob.glob(f'gpurun_out/x25_{v}_{i}/*counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                if 'pop_msym' in r['Kernel_Name']:
                    tot[r['Counter_Name']] += float(r['Counter_Value'])
    print(v, {k: '%.4g' % x for k, x in sorted(tot.items())})
PY
