"""one populations -> neighbours pair on the C3 data (neighbour call with DC_FLAG_STATS_VALID), for kernel traces"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
ct = torch.from_numpy(gaussian_blobs(1_000_000, 10)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
dens.sweep_timing(True)
for rep in range(4):
    ev[0].record(); p = dens.calculate_populations_partial(ct, [0.2]); ev[1].record()
    fe = dens.calculate_free_energies(p[0].contiguous())
    ev[1].record(); nn = dens.nearest_neighbors_partial(ct, fe, stats_valid=True); ev[2].record(); torch.cuda.synchronize()
    print(f"pop call {ev[0].elapsed_time(ev[1]):.2f} (kernel {dens.last_sweep_ms('pop', ct.device):.2f})  nn call {ev[1].elapsed_time(ev[2]):.2f} (kernel {dens.last_sweep_ms('nn', ct.device):.2f})")
