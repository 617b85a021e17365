"""neighbour sweep on data that stresses the folded norms / early-out bound: extents far above the neighbour distances,
lines, duplicates, tiny and huge scales, one dimension dominating"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
rng = np.random.default_rng(11)
def check(c, tag):
    c = np.ascontiguousarray(c, dtype=np.float32)
    ct = torch.from_numpy(c).cuda()
    r = float(np.sqrt(c.shape[1]) * max(c.std(axis=0).mean(), 1e-30))
    p = dens.calculate_populations_partial(ct, [0.3 * r], variant="direct")
    fe = dens.calculate_free_energies(p[0].contiguous())
    want = dens.nearest_neighbors_partial(ct, fe, variant="direct")
    for v in ("pruned", "mfma"):
        got = dens.nearest_neighbors_partial(ct, fe, variant=v)
        for a, b in zip(got, want):
            assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (tag, v)
    assert bool((dens.calculate_populations_partial(ct, [0.3 * r], variant="pruned") == p).all()), (tag, "pop")
    print("ok", tag, c.shape)
for d in (5, 8, 10, 14, 20):
    n = 30000
    tight = rng.normal(0, 1e-3, (n, d))
    out = rng.normal(0, 1.0, (40, d)) * 1e3
    check(np.vstack([tight, out]), f"tight cluster + far outliers d={d}")
    line = np.outer(np.linspace(0, 1, n), np.ones(d)) + rng.normal(0, 1e-6, (n, d))
    check(line, f"line d={d}")
    check(np.repeat(rng.normal(0, 1, (n // 8, d)), 8, axis=0), f"8-fold duplicates d={d}")
    x = rng.normal(0, 1, (n, d)); x[:, 0] *= 1e4
    check(x, f"one column dominating d={d}")
    check(rng.normal(0, 1, (n, d)) * 1e-12 + 1.0, f"tiny spread around 1 d={d}")
    check(rng.normal(0, 1, (n, d)) * 1e12, f"huge scale d={d}")
    two = np.vstack([rng.normal(0, 0.05, (n // 2, d)), rng.normal(0, 0.05, (n // 2, d)) + 7.0])
    check(two, f"two clusters far apart in every column d={d}")
print("adversarial ok")
