#!/bin/bash
# kernel timeline of one C2 step (100k x 10, radii 0.1 0.2 0.3, pop + FE)
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_c2
cat > /tmp/c2step.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(100000, 10)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(4):
    ev[0].record()
    p = dens.calculate_populations_partial(c, [0.1, 0.2, 0.3])
    fe = dens.calculate_free_energies(p[1].contiguous())
    ev[1].record(); torch.cuda.synchronize()
    print("step ms", ev[0].elapsed_time(ev[1]))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_c2 -o s -- python3 /tmp/c2step.py $R 2>/dev/null | tail -2
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_c2/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stats_kernel" in r["Kernel_Name"]]
i0 = idx[-1] - 1
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("dc::(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:5.1f} {name}")
    prev_end = e
PY
python3 /tmp/c2step.py $R 2>/dev/null | tail -2
