"""Neighbour sweep of C3's shape: tile pairs evaluated, MFMAs issued and the kernel's time, for all rows and for one
eighth of the groups -- run under the variants being compared (DC_NN_BOUNDS=0/1, ...)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
p = dens.calculate_populations_partial(c, [0.2])[0]
fe = dens.calculate_free_energies(p.contiguous())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for G, seg in ((8, 3), (1, 0)):
    ts = []
    for _ in range(3):
        ev[0].record(); out = dens.nearest_neighbors_segment(c, fe, seg, G); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    tiles = dens.evaluated_tiles(c.device)[1]; mf = dens.issued_mfmas(c.device)[1]
    print("CNT", json.dumps({"G": G, "call_ms": min(ts), "tile_pairs": tiles, "mfma": mf, "mfma_per_pair": mf / max(tiles, 1),
                             "env": {k: v for k, v in os.environ.items() if k.startswith("DC_NN")}}))
