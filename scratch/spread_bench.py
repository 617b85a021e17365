"""One populations -> free energies -> neighbours step on the C3 blobs with their centres moved apart by a factor
(scratch/spread_exp.py's data), as a small JSON line: call and kernel times, tiles, components."""
import json, sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
f = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n, d, r = 1_000_000, 10, 0.2
base = gaussian_blobs(n, d)
labels = np.random.default_rng(20240).integers(0, 3, n)
centres = np.zeros((3, d), dtype=np.float32)
centres[:, :2] = [(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)]
ct = torch.from_numpy(np.ascontiguousarray(base + (f - 1.0) * centres[labels], dtype=np.float32)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
dens.sweep_timing(True)
rows = []
for rep in range(4):
    ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record()
    info = dens.components_info(ct); tiles_p = dens.evaluated_tiles(ct.device)[0]
    fe = dens.calculate_free_energies(p[0].contiguous())
    ev[2].record(); nn = dens.nearest_neighbors_partial(ct, fe, stats_valid=True); ev[3].record(); torch.cuda.synchronize()
    rows.append((ev[0].elapsed_time(ev[1]), dens.last_sweep_ms("pop", ct.device), ev[2].elapsed_time(ev[3]), dens.last_sweep_ms("nn", ct.device)))
rows = np.array(rows[1:])
print(json.dumps({"workload": f"1M x 10 blobs, centres x {f}, r = {r}", "pop_call_ms": rows[:, 0].min(), "pop_kernel_ms": rows[:, 1].min(),
                  "nn_call_ms": rows[:, 2].min(), "nn_kernel_ms": rows[:, 3].min(), "pop_tiles": tiles_p,
                  "nn_tiles": dens.evaluated_tiles(ct.device)[1], "components": info,
                  "mean_pop": float(p[0].float().mean()), "sigma2": dens.compute_sigma2(nn[1])}))
