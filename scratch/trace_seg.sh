#!/bin/bash
# kernel timeline of ONE rank's step of an 8-way sharded C3 run (last repetition of scratch/seg_bench.py): per kernel
# start offset, duration and the gap to the previous kernel's end
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_seg
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_seg -o s -- python3 $R/scratch/seg_bench.py ${1:-1000000} ${2:-10} ${3:-8} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_seg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last population segment call starts at the last rowstats... find the last 'fine_mark_kernel' (start of the last pop prep)
idx = [i for i, r in enumerate(rows) if "fine_mark_kernel" in r["Kernel_Name"]]
i0 = idx[-1] - 4
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
tot_k = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("dc::(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}")
    prev_end = e; tot_k += e - s
print("span %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, tot_k / 1e3))
PY
