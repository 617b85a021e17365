"""multi-radius shared-operand sweep against the direct kernels (run with DC_POP_SHARED=1)"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(11)
bad = 0
for n, d in [(9000, 30), (5000, 24), (3000, 40), (4097, 27), (33, 30), (20000, 36), (7000, 42)]:
    for n_rad in (2, 3, 4, 5, 8, 9, 17):
        c = gaussian_blobs(n, d, seed=n + d + n_rad)
        c[rng.integers(0, n, n // 5)] = c[rng.integers(0, n, n // 5)]
        ct = torch.from_numpy(c).cuda()
        scale = float(np.sqrt(2 * d)) * 0.08
        radii = [float(x) for x in scale * rng.uniform(0.5, 1.3, n_rad)]
        want = dens.calculate_populations_partial(ct, radii, variant="direct")
        got = dens.calculate_populations_partial(ct, radii, variant="pruned")
        ok = bool((got == want).all())
        lo, hi = n // 4, n // 4 + max(1, n // 2)
        ok2 = bool((dens.calculate_populations_partial(ct, radii, lo, hi, variant="pruned") ==
                    dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")).all())
        acc = torch.zeros_like(want)
        for g in range(3):
            acc += dens.calculate_populations_segment(ct, radii, g, 3)
        ok3 = bool((acc == want).all())
        if not (ok and ok2 and ok3):
            bad += 1
            print("MISMATCH", n, d, n_rad, ok, ok2, ok3)
print("multi-radius check:", "ok" if bad == 0 else f"{bad} mismatches")
# default rule (no forcing needed): rows >= 50000, three or more MFMAs per chain, several radii
if len(sys.argv) > 1 and sys.argv[1] == "auto":
    for n, d, n_rad in [(60000, 16, 5), (80000, 30, 2), (50000, 12, 3), (70000, 40, 8)]:
        c = gaussian_blobs(n, d, seed=n + d)
        ct = torch.from_numpy(c).cuda()
        radii = [float(x) for x in float(np.sqrt(2 * d)) * 0.08 * rng.uniform(0.5, 1.1, n_rad)]
        want = dens.calculate_populations_partial(ct, radii, variant="direct")
        assert bool((dens.calculate_populations_partial(ct, radii) == want).all()), (n, d, n_rad)
        acc = torch.zeros_like(want)
        for g in range(4):
            acc += dens.calculate_populations_segment(ct, radii, g, 4)
        assert bool((acc == want).all()), (n, d, n_rad, "segments")
    print("auto rule: ok")
