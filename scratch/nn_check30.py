"""bit-for-bit check of the shared-operand neighbour sweep (DC_NN_SHARED=1 forces it) against the direct kernels:
n_cols 27 / 30 / 31 (NM = 6, two or three MFMAs in front of the early-out), all rows, a row range, segments; duplicates"""
import os, sys
os.environ["DC_NN_SHARED"] = "1"
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(7)
for n, d, r in [(30000, 30, 0.5), (9000, 27, 0.45), (20000, 31, 0.5), (700, 30, 0.5)]:
    c = gaussian_blobs(n, d, seed=3 * n + d)
    if n > 100:
        c[rng.integers(0, n, n // 6)] = c[rng.integers(0, n, n // 6)]
    ct = torch.from_numpy(c).cuda()
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, [r], variant="direct")[0].contiguous())
    want = dens.nearest_neighbors_partial(ct, fe, variant="direct")
    got = dens.nearest_neighbors_partial(ct, fe, variant="pruned")
    for a, b in zip(got, want):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "all rows")
    lo, hi = n // 3, n // 3 + max(1, n // 2)
    for a, b in zip(dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="pruned"), dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "row range")
    words = None
    for g in range(3):
        w = dens.pack_neighbors(*dens.nearest_neighbors_segment(ct, fe, g, 3))
        words = w if words is None else torch.minimum(words, w)
    for a, b in zip(dens.unpack_neighbors(words), want):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "segments")
print("nn_check30 ok")
