import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = int(sys.argv[1]), int(sys.argv[2])
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
ch = gaussian_blobs(n, d)
c = torch.from_numpy(ch).cuda()
pops = dens.calculate_populations_partial(c, [radius])
fe = dens.calculate_free_energies(pops[0].contiguous())
nn = dens.nearest_neighbors_partial(c, fe)
sigma2 = dens.compute_sigma2(nn[1])
r2 = np.float32(4 * sigma2)
order = np.argsort(fe.cpu().numpy(), kind="stable")
rank = np.empty(n, dtype=np.uint32)
rank[order] = np.arange(n, dtype=np.uint32)
torch.cuda.synchronize()
# one round on device-resident data
comp = torch.arange(n, dtype=torch.int32, device="cuda")
rk = torch.from_numpy(rank.astype(np.int32)).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    best, p2 = dens.radius_min_edge(c, r2, comp, rk)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"one min-edge round (singletons): {1e3*(t1-t0):.1f} ms; mean partners {float(p2.float().mean())-1:.1f}")
for rep in range(2):
    t0 = time.time()
    edges, rounds = dens.radius_forest(ch, r2, rank)
    t1 = time.time()
    print(f"n={n} d={d} sigma2={sigma2:.6g} r2={r2:.6g}: forest of {len(edges)} pairs ({n-len(edges)} components) in {rounds} sweeps, {1e3*(t1-t0):.1f} ms incl. upload")
