"""probe builds only: share of a phase in the summed wave time of the neighbour sweep (ticks of 10 ns)"""
import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(1000000, 10)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
dens.nearest_neighbors_partial(c, fe)
phase, total = dens.evaluated_tiles(c.device)
print(f"{sys.argv[1]}: {100.0 * phase / total:.1f} % of the summed wave time ({total / 1e8:.2f} wave-seconds in total)")
