import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, G = 1000000, 10, int(sys.argv[1])
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
for seg in range(G):
    dens.nearest_neighbors_segment(c, fe, seg, G)
    v = dens.evaluated_tiles(c.device)[1]
    print(f"segment {seg}: slowest wave {(v >> 40) / 100.0:.1f} us, chains {(v >> 20) & 0xFFFFF}, group {v & 0xFFFFF}")
