"""Unfavourable workloads for the pruned sweeps (the headline leans on three well-separated blobs):
one populations -> free energies -> neighbours step at 1M x 10, r = 0.2, on
  oneblob   ONE Gaussian blob (sigma 0.08) of all 10^6 frames: every pair is an intra-cluster pair, three times C3's
  uniform   frames uniform in the unit box: no pair within r, the neighbour sweep's 2-D boxes bound almost nothing
as a small JSON line: call and kernel times, evaluated tile pairs and fractions."""
import json, sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
kind = sys.argv[1] if len(sys.argv) > 1 else "oneblob"
n, d, r = 1_000_000, 10, 0.2
rng = np.random.default_rng(20240)
c = (rng.normal(0.0, 0.08, (n, d)) if kind == "oneblob" else rng.uniform(0.0, 1.0, (n, d))).astype(np.float32)
ct = torch.from_numpy(c).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
dens.sweep_timing(True)
rows = []
for rep in range(3):
    ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record()
    tiles_p = dens.evaluated_tiles(ct.device)[0]
    fe = dens.calculate_free_energies(p[0].contiguous())
    ev[2].record(); nn = dens.nearest_neighbors_partial(ct, fe, stats_valid=True); ev[3].record(); torch.cuda.synchronize()
    rows.append((ev[0].elapsed_time(ev[1]), dens.last_sweep_ms("pop", ct.device), ev[2].elapsed_time(ev[3]), dens.last_sweep_ms("nn", ct.device)))
rows = np.array(rows[1:])
tiles_n = dens.evaluated_tiles(ct.device)[1]
T = (n + 31) // 32
step = rows[:, 0].min() + rows[:, 2].min()
print(json.dumps({"workload": f"1M x 10, {kind}, r = {r}", "pop_call_ms": rows[:, 0].min(), "pop_kernel_ms": rows[:, 1].min(),
                  "nn_call_ms": rows[:, 2].min(), "nn_kernel_ms": rows[:, 3].min(), "pop_tiles": tiles_p, "nn_tiles": tiles_n,
                  "evaluated_fraction": {"pop (computed tile pairs x 2, symmetric)": 2.0 * tiles_p / (T * T), "nn": tiles_n / (T * T)},
                  "frame_pairs_per_s": 2.0 * n * n / (step * 1e-3), "components": dens.components_info(ct),
                  "mean_pop": float(p[0].float().mean()), "sigma2": dens.compute_sigma2(nn[1])}))
