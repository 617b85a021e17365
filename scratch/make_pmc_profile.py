#!/usr/bin/env python3
"""profiles/r1_pruned_pmc.json from rocprofv3 --pmc passes (gpurun_out/pmc_f1, pmc_f2: SQ sets; pmc_f3: TCC_EA0)."""
import csv, json, subprocess, sys
sq = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py', 'gpurun_out/pmc_f1', 'gpurun_out/pmc_f2']))
mem = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py', 'gpurun_out/pmc_f3']))
bench = json.load(open('profiles/r1_pruned_bench.json'))
frac = bench['roofline']['evaluated_fraction']
dur = {}
for r in csv.DictReader(open('gpurun_out/pmc_f1/s_kernel_trace.csv')):
    k = r['Kernel_Name'][:64]
    if k in sq:
        dur.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
chains = {'pop': frac['pop'] * (31250 ** 2), 'nn': frac['nn'] * (31250 ** 2)}
out = {"note": "rocprofv3 --kernel-trace --pmc, counters in their own passes (two passes for the SQ sets, one for the two "
               "TCC_EA0 request counters; FETCH_SIZE hung the profiler on this kernel and is not used), on "
               "`scratch/kbench.py --n 1000000 --d 10 --variant pruned --reps 1` (C3: 1M x 10, r = 0.2). Values are per "
               "dispatch, summed over XCDs/SEs as rocprofv3 reports them. traffic_bytes = TCC_EA0_RDREQ_sum * 128 B + "
               "TCC_EA0_WRREQ_sum * 64 B: on gfx950 the memory-side read requests of 16-byte-per-lane streaming loads are "
               "128-B requests tallied at 64 B (MI355X_MICROARCH.md, HBM section: double FETCH_SIZE = RDREQ x 64 B); "
               "Infinity-Cache hits are included, so this is an upper bound of the HBM bytes. clock_ghz = GRBM_GUI_ACTIVE / "
               "8 XCDs / duration; matrix_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); "
               "valu_insts_per_32x32_tile_pair divides by the evaluated tile pairs of profiles/r1_pruned_bench.json.",
       "workload": {"n_rows": 1000000, "n_cols": 10, "radii": [0.2], "variant": "pruned"}, "kernels": {}}
for k in sq:
    e = dict(sq[k])
    mk = [m for m in mem if m[:50] == k[:50]]
    if mk:
        e.update({c: v for c, v in mem[mk[0]].items() if c != 'dispatches'})
    key = 'pop' if 'pop_' in k else 'nn'
    d = min(dur[k]); cyc = e['GRBM_GUI_ACTIVE'] / 8
    e['duration_ms_under_counters'] = d * 1e3
    e['clock_ghz'] = cyc / d / 1e9
    e['matrix_pipe_utilisation'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
    e['valu_insts_per_32x32_tile_pair'] = e['SQ_INSTS_VALU'] / chains[key]
    if 'TCC_EA0_RDREQ_sum' in e:
        e['traffic_bytes'] = e['TCC_EA0_RDREQ_sum'] * 128 + e['TCC_EA0_WRREQ_sum'] * 64
    out['kernels'][k] = e
json.dump(out, open('profiles/r1_pruned_pmc.json', 'w'), indent=1)
for k, e in out['kernels'].items():
    print(k[-45:], {x: round(e[x], 3) for x in ['duration_ms_under_counters', 'clock_ghz', 'matrix_pipe_utilisation', 'valu_insts_per_32x32_tile_pair'] if x in e}, e.get('traffic_bytes'))
