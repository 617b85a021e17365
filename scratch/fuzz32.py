"""Random shapes / row ranges / radii / scales at 9 - 10 columns: the fp32-input MFMA variant (dc_mfma32.hpp: scaled image,
two-bit epilogue, reference chunks, parked candidates) against the direct kernels.  DC_MFMA32_CHUNKS forces a chunk count."""
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
    n = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000, 3000, 9000, 40000, 100000], p=[.03,.03,.05,.05,.05,.05,.1,.14,.2,.12,.1,.06,.02]))
    d = int(rng.choice([9, 10]))
    kind = int(rng.integers(0, 8))
    c = gaussian_blobs(n, d, seed=int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.02, 0.08, 0.3])))
    if kind == 1:   # duplicates
        c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
    if kind == 2:   # large offset (cancellation stress)
        c += np.float32(rng.choice([10.0, 1000.0]))
    if kind == 3:
        c *= np.float32(1e-3)
    if kind == 4:   # far from 1: the scale of the population image
        c *= np.float32(rng.choice([1e-12, 1e-6, 1e4, 1e8, 1e15]))
    if kind == 5:   # a lattice: massive ties and pairs exactly on the radius
        c = (rng.integers(0, 4, size=(n, d)) * 0.25).astype(np.float32)
    if kind == 6 and n > 1:   # all rows equal but one
        c[:] = c[0]
        c[n // 2] += np.float32(0.5)
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    scale = float(np.sqrt(d)) * float(c.std(axis=0).mean() if n > 1 else 1.0)
    if not scale > 0: scale = 1.0
    radii = [float(x) for x in (scale * rng.uniform(0.05, 1.5, size=int(rng.choice([1, 1, 2, 3, 5]))))]
    if kind == 5: radii = [0.25, 0.5, float(np.sqrt(np.float32(0.125))), 0.75][:len(radii) + 1]
    if rng.random() < 0.1: radii[0] = 0.0
    if rng.random() < 0.05: radii[-1] = 1e18
    lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
    if rng.random() < 0.5: lo, hi = 0, n
    ref_p = dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, radii[:1], variant="direct")[0].contiguous())
    ref_n = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")
    p = dens.calculate_populations_partial(ct, radii, lo, hi, variant="mfma32")
    q = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="mfma32")
    okp = bool((p == ref_p).all())
    okn = all(bool((x.view(torch.int32) == y.view(torch.int32)).all()) for x, y in zip(q, ref_n))
    if not (okp and okn):
        bad += 1
        print(f"MISMATCH case {case}: n={n} d={d} kind={kind} radii={radii} rows=[{lo},{hi}) pops_ok={okp} nn_ok={okn}")
print(f"{n_cases} cases, {bad} mismatches, {time.time()-t0:.1f}s")
