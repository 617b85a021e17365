"""How the guard band scales with the spread of the data: the C3 blobs (sigma 0.08, r = 0.2) with their centres moved
apart by a factor f (max |x'|^2 of the centred data grows with f^2, and with it the band of the Gram form)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
base = gaussian_blobs(n, d)
rng = np.random.default_rng(20240)
labels = rng.integers(0, 3, n)                       # the generator's own label draw (first draw of the seed)
centres = np.zeros((3, d), dtype=np.float32)
centres[:, :2] = [(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for f in (1.0, 3.0, 10.0, 30.0):
    c = base + (f - 1.0) * centres[labels]
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    ts = []
    for rep in range(3):
        ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    tiles = dens.evaluated_tiles(ct.device)[0]
    fe = dens.calculate_free_energies(p[0].contiguous())
    tn = []
    for rep in range(2):
        ev[0].record(); dens.nearest_neighbors_partial(ct, fe); ev[1].record(); torch.cuda.synchronize()
        tn.append(ev[0].elapsed_time(ev[1]))
    print(f"centres x {f:4.0f}: pop {min(ts):7.2f} ms ({tiles*1024/n/n:.3f} of the pairs evaluated), nn {min(tn):7.2f} ms, mean pop {float(p[0].float().mean()):.1f}")
