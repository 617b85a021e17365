import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(1_000_000, 10)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
for rep in range(2):
    dens.nearest_neighbors_partial(c, fe, stats_valid=True)
torch.cuda.synchronize()
h = dens._workspace(c.device).buf[:128].view(torch.int32).cpu().tolist()
print("open queries listed:", h[30])
