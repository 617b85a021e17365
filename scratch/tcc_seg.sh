#!/bin/bash
# memory-side read requests of the pruned sweeps, unsharded and summed over the eight segments
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
for G in 1 8; do
  rm -rf $R/gpurun_out/tcc_g$G
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/tcc_g$G -o s -- python3 $R/scratch/seg_bench.py 1000000 10 $G > /dev/null 2>&1
  (cd $R; python3 - <<P
import csv,glob,collections
f=glob.glob('gpurun_out/tcc_g$G/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'pruned_kernel' not in k: continue
    k=k[30:52]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[k]+=1
for k,v in acc.items():
    print('G=$G',k,'dispatches',n[k],'read GB per dispatch',round(v['TCC_EA0_RDREQ_sum']*128/1e9/n[k],3),'total GB',round(v['TCC_EA0_RDREQ_sum']*128/1e9,2),'L2 hit rate',round(v['TCC_HIT_sum']/max(1,v['TCC_HIT_sum']+v['TCC_MISS_sum']),3))
P
)
done
