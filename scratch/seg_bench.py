"""Time of one segment of a G-way sharded run (what one rank of bench.py --gpus G computes), per phase: the calls
ShardedDensity makes (populations segment, free energies, neighbours segment with DC_FLAG_STATS_VALID, block pack),
each split into the sweep kernel (library event pair) and the rest of the call (prep)."""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
dens.sweep_timing(True)
rows = []
for seg in list(range(G)) * 2:
    ev[0].record(); dens.calculate_populations_segment(c, [0.2], seg, G)
    ev[1].record(); dens.calculate_free_energies(pops[0].contiguous())
    ev[2].record(); nn = dens.nearest_neighbors_segment(c, fe, seg, G, stats_valid=True)
    ev[3].record(); dens.pack_neighbor_block(c, *nn, seg, G)
    ev[4].record(); torch.cuda.synchronize()
    pk, nk = dens.last_sweep_ms("pop", c.device), dens.last_sweep_ms("nn", c.device)
    rows.append((ev[0].elapsed_time(ev[1]), pk, ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), nk, ev[3].elapsed_time(ev[4])))
r = np.array(rows[G:])
pop, popk, fe_t, nnc, nnk, pack = r.T
out = {"n_rows": n, "n_cols": d, "segments": G,
       "pop_call_ms": {"mean": pop.mean(), "max": pop.max()}, "pop_kernel_ms": {"mean": popk.mean(), "max": popk.max()},
       "fe_ms": fe_t.mean(), "nn_call_ms": {"mean": nnc.mean(), "max": nnc.max()},
       "nn_kernel_ms": {"mean": nnk.mean(), "max": nnk.max()}, "block_pack_ms": pack.mean(),
       "prep_ms": float((pop - popk).mean() + (nnc - nnk).mean()),
       "per_rank_pop_nn_prep_ms": float(pop.max() + nnc.max()),
       "per_rank_step_ms_before_collectives": float(pop.max() + fe_t.mean() + nnc.max() + pack.mean())}
print("   nn per segment:", np.round(nnc, 2), " pop:", np.round(pop, 2))
print("SEG " + json.dumps(out))
