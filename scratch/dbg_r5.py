import sys, numpy as np, ctypes as C
import torch
sys.path.insert(0, '.')
from clustering_amd import capi, density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
import torch
o = Oracle()
c = gaussian_blobs(2000, 10, seed=3)
n, d = c.shape
radii = [0.2, 0.1]
full = o.populations(c, radii)
fe = o.free_energies(full[0])
exp = o.nearest_neighbors(c, fe)
ct = torch.from_numpy(c).cuda()
for sv in (False, True):
    pops = dens.calculate_populations_partial(ct, radii)
    print("pops ok", bool((pops.cpu().numpy().astype(np.uint64) == full).all()))
    fet = dens.calculate_free_energies(pops[0].contiguous())
    nn = dens.nearest_neighbors_partial(ct, fet, stats_valid=sv)
    got = [t.cpu().numpy() for t in nn]
    bad0 = np.nonzero(got[0].astype(np.uint64) != exp[0])[0]
    bad2 = np.nonzero(got[2].astype(np.uint64) != exp[2])[0]
    print("stats_valid", sv, "nn mismatches", len(bad0), "hd mismatches", len(bad2))
    for i in list(bad0[:5]) + list(bad2[:5]):
        print("  row", i, "got", got[0][i], got[1][i], got[2][i], got[3][i], "want", exp[0][i], exp[1][i], exp[2][i], exp[3][i], "fe", fe[i])
    print("  comps", dens.components_info(ct))
