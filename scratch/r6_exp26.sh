#!/bin/bash
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out
rm -rf $O/x26
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/x26 -o s -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-full-sweep > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(collections.Counter); n = collections.Counter()
for f in glob.glob('gpurun_out/x26/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in tot.items():
    if c['SQ_INSTS_VALU'] > 1e9: print(k, {a: '%.4g' % b for a, b in sorted(c.items())})
PY
