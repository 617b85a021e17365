"""probe build only (scratch/build_variant.sh with the counters patched in): slowest wave of every segment"""
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
    w, v = dens.evaluated_tiles(c.device)
    print(f"segment {seg}: slowest wave {(v >> 44) / 100.0:.1f} us, chains {(v >> 28) & 0xFFFF}, rare {(v >> 14) & 0x3FFF}, special {v & 0x3FFF} | trig {(w >> 30) & 0x3FFF}, flush slots {(w >> 14) & 0xFFFF}, rings {(w >> 6) & 0xFF}, chunk {w & 0x3F}")
