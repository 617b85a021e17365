#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel (averaged per dispatch).
usage: pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout"""
import csv, glob, json, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:64]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k, cs in tot.items():
    if "pruned_kernel" not in k and "mfma_kernel" not in k:
        continue
    out[k] = {c: v / max(1, len(disp[(k, c)])) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(disp[(k, c)]) for c in cs)
print(json.dumps(out, indent=1))
