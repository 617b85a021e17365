"""Tile pairs evaluated by the segments of a G-way sharded run against the unsharded sweep (the cost of splitting the
reference axis further: every share confirms its rings with its own incumbents)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
dens.sweep_timing(True)
for G in (1, 2, 4, 8):
    tp = tn = 0; kp = kn = 0.0
    for seg in range(G):
        dens.calculate_populations_segment(c, [0.2], seg, G); torch.cuda.synchronize()
        tp += dens.evaluated_tiles(c.device)[0]; kp += dens.last_sweep_ms("pop", c.device)
        dens.nearest_neighbors_segment(c, fe, seg, G, stats_valid=True); torch.cuda.synchronize()
        tn += dens.evaluated_tiles(c.device)[1]; kn += dens.last_sweep_ms("nn", c.device)
    print(f"G={G}: pop tiles {tp/1e6:.1f}M kernel sum {kp:.2f} ms ({kp/G:.3f} per segment) | nn tiles {tn/1e6:.1f}M kernel sum {kn:.2f} ms ({kn/G:.3f} per segment)")
