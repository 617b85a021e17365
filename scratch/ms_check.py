"""quick bit-for-bit check of the multi-radius symmetric sweep (D = 30 only: developer builds with MFMA_STEPS=6) against
the direct kernels: all rows and the sum over segments, 3 / 4 / 8 radii, duplicates"""
import os, sys
os.environ.setdefault("DC_POP_SHARED", "1")
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(3)
for n, d, r in [(9000, 30, 0.5), (70000, 30, 0.55), (33, 30, 0.5), (4097, 28, 0.5)]:
    c = gaussian_blobs(n, d, seed=n + d)
    if n > 100:
        c[rng.integers(0, n, n // 7)] = c[rng.integers(0, n, n // 7)]
    ct = torch.from_numpy(c).cuda()
    for n_rad, order in ((3, "any"), (4, "any"), (8, "any"), (8, "ascending"), (4, "ascending"), (5, "ascending"), (8, "wide")):
        radii = [float(x) for x in r * rng.uniform(0.6, 1.25, n_rad)]
        if order == "ascending":   # (the radii a chain holds nothing of are skipped: dc_mfma_msym.hpp mr_chain_k)
            radii = sorted(radii)
        if order == "wide":        # from far below to far above the typical pair distance: every skip count occurs
            radii = [float(x) for x in r * np.linspace(0.25, 1.6, n_rad)]
        want = dens.calculate_populations_partial(ct, radii, variant="direct")
        got = dens.calculate_populations_partial(ct, radii, variant="pruned")
        assert bool((got == want).all()), (n, d, n_rad, int((got != want).sum()))
        acc = torch.zeros_like(want)
        for g in range(3):
            acc += dens.calculate_populations_segment(ct, radii, g, 3)
        assert bool((acc == want).all()), (n, d, n_rad, "segments", int((acc != want).sum()))
print("ms_check ok")
