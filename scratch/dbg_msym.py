import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(7)
for n, d, r, radii in [(9000, 30, 0.5, [0.5, 0.4]), (9000, 30, 0.5, [0.4, 0.5]), (9000, 30, 0.5, [0.5, 0.4, 0.45, 0.55]), (9000, 30, 0.5, [0.3, 0.4, 0.5, 0.6, 0.7]), (9000, 30, 0.5, [0.5])]:
    c = gaussian_blobs(n, d, seed=n + d)
    ct = torch.from_numpy(c).cuda()
    want = dens.calculate_populations_partial(ct, radii, variant="direct")
    got = dens.calculate_populations_partial(ct, radii, variant="pruned")
    for k in range(len(radii)):
        dif = (got[k].to(torch.int64) - want[k].to(torch.int64))
        print(radii, 'radius', k, 'rows differing', int((dif != 0).sum()), 'sum diff', int(dif.sum()), 'max', int(dif.abs().max()), 'mean pop', float(want[k].float().mean()))
