import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(1000000, 10)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
p0 = pops[0].contiguous()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fe = dens.calculate_free_energies(p0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"free energies: {1e3*(t1-t0):.3f} ms")
