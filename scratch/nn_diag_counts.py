import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n = 1000000
c = torch.from_numpy(gaussian_blobs(n, 10)).cuda()
p = dens.calculate_populations_partial(c, [0.2], variant="pruned")
fe = dens.calculate_free_energies(p[0].contiguous())
dens.nearest_neighbors_partial(c, fe, variant="pruned")
torch.cuda.synchronize()
print("counters (B, A):", dens.evaluated_tiles(c.device))
