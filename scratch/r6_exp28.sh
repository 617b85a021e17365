#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out
export DC_LIB_PATH=$R/clustering_amd/lib/variants/r6_$1.so
rm -rf $O/x28
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/x28 -o s -- python3 $R/scratch/c5_bench.py --reps 1 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(collections.Counter); n=collections.Counter()
for f in glob.glob('gpurun_out/x28/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][32:72]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k, c in tot.items():
    if c['SQ_INSTS_VALU'] > 1e9: print(k, n[k], 'valu %.3g br %.3g salu %.3g lds %.3g br/valu %.3f wait %.2f waitinst %.2f active %.2f' % (c['SQ_INSTS_VALU'], c['SQ_INSTS_BRANCH'], c['SQ_INSTS_SALU'], c['SQ_INSTS_LDS'], c['SQ_INSTS_BRANCH']/c['SQ_INSTS_VALU'], c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']))
PY
