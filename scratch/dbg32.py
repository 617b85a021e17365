import numpy as np, torch, sys
sys.path.insert(0,".")
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
for n in (128, 256, 3000):
    c = gaussian_blobs(n, 10, seed=7000)
    ct = torch.from_numpy(c).cuda()
    for r in (0.2, 5.0, 0.01, 0.5):
        a = dens.calculate_populations_partial(ct, [r], variant="mfma32").cpu().numpy()[0]
        b = dens.calculate_populations_partial(ct, [r], variant="direct").cpu().numpy()[0]
        d = a.astype(np.int64)-b
        nz = np.nonzero(d)[0]
        print(n, r, "mismatch", len(nz), "sum diff", d.sum(), "tiles mod 4:", np.bincount((nz//32) % 4, minlength=4), nz[:8], d[nz[:8]])
