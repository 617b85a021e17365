#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel (averaged per dispatch).
usage: pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout"""
import csv, glob, json, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
rows = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
# a kernel launched with several grid sizes (a full sweep and one segment of it) is reported per grid size
grids = collections.defaultdict(set)
for r in rows:
    grids[r["Kernel_Name"][:64]].add(r["Grid_Size"])
for _ in [0]:
    for _ in [0]:
        for r in rows:
            k = r["Kernel_Name"][:64]
            if len(grids[k]) > 1:
                k += " grid=" + r["Grid_Size"]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k, cs in tot.items():
    if "pruned_kernel" not in k and "mfma_kernel" not in k and "shared_kernel" not in k and "msym_kernel" not in k and "mfma32_kernel" not in k:
        continue
    out[k] = {c: v / max(1, len(disp[(k, c)])) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(disp[(k, c)]) for c in cs)
print(json.dumps(out, indent=1))
