import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = int(sys.argv[1]), int(sys.argv[2])
lo, hi = int(sys.argv[3]), int(sys.argv[4])
radii = [float(x) for x in sys.argv[5:]] or [0.5]
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
t0 = time.time()
pp = dens.calculate_populations_partial(c, radii, lo, hi, variant="pruned"); torch.cuda.synchronize()
t1 = time.time()
pd = dens.calculate_populations_partial(c, radii, lo, hi, variant="direct"); torch.cuda.synchronize()
t2 = time.time()
print(f"pops pruned {t1-t0:.2f}s direct {t2-t1:.2f}s equal={bool((pp == pd).all())} mean={pp[0, lo:hi].float().mean().item():.1f}")
# free energies from a full-range population (pruned), then neighbours on the row range with both variants
pf = dens.calculate_populations_partial(c, radii[:1], variant="pruned")
fe = dens.calculate_free_energies(pf[0].contiguous())
torch.cuda.synchronize(); t3 = time.time()
a = dens.nearest_neighbors_partial(c, fe, lo, hi, variant="pruned"); torch.cuda.synchronize(); t4 = time.time()
b = dens.nearest_neighbors_partial(c, fe, lo, hi, variant="direct"); torch.cuda.synchronize(); t5 = time.time()
ok = all(bool((x[lo:hi] == y[lo:hi]).all()) for x, y in zip(a, b))
print(f"nn pruned {t4-t3:.2f}s direct {t5-t4:.2f}s equal={ok}  full pop sweep {t3-t2:.2f}s")
