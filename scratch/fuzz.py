"""Random shapes / row ranges / radii: the matrix-core variants against the direct kernels (all product code)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000, 3000, 9000, 40000, 150000], p=[.03,.03,.05,.05,.05,.05,.1,.14,.2,.12,.08,.07,.03]))
    d = int(rng.integers(1, 33))
    kind = rng.integers(0, 4)
    c = gaussian_blobs(n, d, seed=int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.02, 0.08, 0.3])))
    if kind == 1:   # duplicates
        c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
    if kind == 2:   # large offset (cancellation stress)
        c += np.float32(rng.choice([10.0, 1000.0]))
    if kind == 3:   # tiny scale
        c *= np.float32(1e-3)
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    scale = float(np.sqrt(d)) * float(c.std(axis=0).mean() if n > 1 else 1.0)
    radii = [float(x) for x in (scale * rng.uniform(0.05, 1.5, size=int(rng.integers(1, 4))))]
    lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
    if rng.random() < 0.5: lo, hi = 0, n
    ref_p = dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, radii[:1], variant="direct")[0].contiguous())
    ref_n = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")
    for v in ("pruned", "mfma"):
        p = dens.calculate_populations_partial(ct, radii, lo, hi, variant=v)
        q = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant=v)
        ok = bool((p == ref_p).all()) and all(bool((x.view(torch.int32) == y.view(torch.int32)).all()) for x, y in zip(q, ref_n))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} variant {v}: n={n} d={d} kind={kind} radii={radii} rows=[{lo},{hi}) pops_ok={bool((p == ref_p).all())}")
print(f"{n_cases} cases, {bad} mismatches, {time.time()-t0:.1f}s")
