"""C5 shape (5M x 30, 8 radii): what one rank of an 8-GPU run computes (segment 3 of 8), timed per phase."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, G = 5_000_000, 30, 8
radii = [0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 0.65]
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for rep in range(2):
    ev[0].record()
    p = dens.calculate_populations_segment(c, radii, 3, G)
    ev[1].record(); torch.cuda.synchronize()
    print(f"pops, 8 radii, segment 3/8: {ev[0].elapsed_time(ev[1]):.1f} ms")
# FE needs the populations of all rows: one full single-radius sweep here (a real run all-reduces the segments)
t0 = time.time(); pf = dens.calculate_populations_partial(c, [0.5]); torch.cuda.synchronize(); print(f"full single-radius sweep: {1e3*(time.time()-t0):.1f} ms")
fe = dens.calculate_free_energies(pf[0].contiguous())
for rep in range(2):
    ev[0].record(); nn = dens.nearest_neighbors_segment(c, fe, 3, G); ev[1].record(); torch.cuda.synchronize()
    print(f"nn segment 3/8: {ev[0].elapsed_time(ev[1]):.1f} ms; evaluated tiles {dens.evaluated_tiles(c.device)[1]}")
