import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = int(sys.argv[1]), int(sys.argv[2])
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
nn = dens.nearest_neighbors_partial(c, fe)
sigma2 = dens.compute_sigma2(nn[1])
r2 = np.float32(4 * sigma2)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time()
    pairs, p2 = dens.radius_pairs(c, r2)
    torch.cuda.synchronize()
    t1 = time.time()
    print(f"sigma2={sigma2:.6g} r2={r2:.6g} pairs={pairs.shape[0]} mean partners={2*pairs.shape[0]/n:.2f} time {1e3*(t1-t0):.1f} ms (count + list sweeps)")
