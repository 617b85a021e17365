#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from rocprofv3 --pmc passes.
usage: make_pmc_profile.py <tag> <bench.json> <workload-json> <sq dirs...> -- <tcc dirs...>
  bench.json: a bench.py line (roofline.evaluated_fraction + config) or a scratch/c5_bench.py line (roofline_pop/nn.tile_pairs)."""
import csv, glob, json, subprocess, sys
tag, bench_path, workload = sys.argv[1], sys.argv[2], json.loads(sys.argv[3])
rest = sys.argv[4:]
sq_dirs, tcc_dirs = rest[:rest.index('--')], rest[rest.index('--') + 1:]
sq = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py'] + sq_dirs)) if sq_dirs else {}
mem = json.loads(subprocess.check_output(['python3', 'scratch/pmc_summary.py'] + tcc_dirs)) if tcc_dirs else {}
bench = json.loads(open(bench_path).read().strip().splitlines()[-1])
if 'roofline_pop' in bench:
    chains = {'pop': bench['roofline_pop']['tile_pairs'], 'nn': bench['roofline_nn']['tile_pairs']}
    # (scratch/c5_bench.py also runs ONE full one-radius population sweep for the free energies -- another kernel, its own count)
    if 'full sweep' in bench.get('evaluated_fraction', {}):
        chains['pop_full'] = bench['evaluated_fraction']['full sweep'] * float(workload['n_rows']) ** 2 / 1024.0
elif 'pop_tiles' in bench:   # scratch/spread_bench.py
    chains = {'pop': bench['pop_tiles'], 'nn': bench['nn_tiles']}
else:
    fr = bench['roofline']['evaluated_fraction']
    t = (bench['config']['n_rows'] + 31) // 32
    chains = {'pop': fr['pop'] * t * t, 'nn': fr['nn'] * t * t}   # (the counter sums over the radii of a call)
dur = {}
seen = set()
for d in sq_dirs + tcc_dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        grids = {}
        for r in rows:
            grids.setdefault(r['Kernel_Name'][:64], set()).add(r['Grid_Size'])
        for r in rows:
            k = r['Kernel_Name'][:64]
            if len(grids[k]) > 1:
                k += ' grid=' + r['Grid_Size']
            if (f, r['Dispatch_Id']) in seen:
                continue
            seen.add((f, r['Dispatch_Id']))
            dur.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
out = {"note": "rocprofv3 --kernel-trace --pmc, counters in their own passes (SQ sets and the two TCC_EA0 request counters "
               "separately). Values are per dispatch, summed over XCDs/SEs as rocprofv3 reports them. traffic_bytes = "
               "TCC_EA0_RDREQ_sum * 128 B + TCC_EA0_WRREQ_sum * 64 B (gfx950: the memory-side read requests of 16-byte-per-lane "
               "streaming loads are 128-B requests, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included, so "
               "this is an upper bound of the HBM bytes. clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration; "
               "matrix_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); "
               "valu_insts_per_32x32_tile_pair divides by the evaluated tile pairs the kernels counted themselves. "
               "A population call sweeps once per radius: per-dispatch values are per radius.",
       "workload": workload, "csrc_digest": (next((open(f).read().strip() for f in ('gpurun_out/' + tag.split('_')[0] + '_csrc_digest.txt', 'gpurun_out/r4_csrc_digest.txt') if __import__('os').path.exists(f)), None)),
       "commit": subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None,
       "kernels": {}}
for k in set(sq) | set(mem):
    e = dict(sq.get(k, {}))
    mk = [m for m in mem if m == k] or [m for m in mem if m[:50] == k[:50]]
    if mk:
        e.update({c: v for c, v in mem[mk[0]].items() if c != 'dispatches' or 'dispatches' not in e})
    key = 'pop' if 'pop_' in k else 'nn'
    if key == 'pop' and 'pop_full' in chains and 'msym' not in k:
        key = 'pop_full'
    ds = dur.get(k, [])
    if ds:
        e['duration_ms_under_counters'] = 1e3 * sum(ds) / len(ds)
    if 'GRBM_GUI_ACTIVE' in e and ds:
        cyc = e['GRBM_GUI_ACTIVE'] / 8
        e['clock_ghz'] = cyc / (sum(ds) / len(ds)) / 1e9
        e['matrix_pipe_utilisation'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
        n_disp = workload.get('dispatches_per_call', {}).get(key, 1)
        # (the unpruned sweeps -- pop_mfma_kernel / nn_mfma_kernel / *_mfma32_kernel -- evaluate EVERY tile pair: ceil(N/32)^2,
        #  not the pruned sweeps' count: round 5 divided them by the wrong one, 7x too high)
        unpruned = ('_mfma_kernel' in k) or ('_mfma32_kernel' in k)
        t_all = ((workload['n_rows'] + 31) // 32) ** 2
        e['tile_pairs_per_dispatch'] = t_all if unpruned else chains[key] / n_disp
        e['valu_insts_per_32x32_tile_pair'] = e['SQ_INSTS_VALU'] / e['tile_pairs_per_dispatch']
    if 'TCC_EA0_RDREQ_sum' in e:
        e['traffic_bytes'] = e['TCC_EA0_RDREQ_sum'] * 128 + e['TCC_EA0_WRREQ_sum'] * 64
        if ds:
            e['traffic_tb_per_s'] = e['traffic_bytes'] / (sum(ds) / len(ds)) / 1e12
    out['kernels'][k] = e
json.dump(out, open(f'profiles/{tag}_pmc.json', 'w'), indent=1)
for k, e in out['kernels'].items():
    print(k[-45:], {x: round(e[x], 3) for x in ['duration_ms_under_counters', 'clock_ghz', 'matrix_pipe_utilisation', 'valu_insts_per_32x32_tile_pair', 'traffic_tb_per_s'] if x in e}, e.get('traffic_bytes'))
