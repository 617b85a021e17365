"""Random shapes / row ranges / radii: the matrix-core variants against the direct kernels (all product code)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
big = len(sys.argv) > 3 and sys.argv[3] == 'big'   # large shapes, clustered data only
bad = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000, 3000, 9000, 40000, 150000], p=[.03,.03,.05,.05,.05,.05,.1,.14,.2,.12,.08,.07,.03]))
    d = int(rng.integers(1, 65)) if rng.random() < 0.3 else int(rng.integers(1, 33))
    if rng.random() < 0.15: d = int(rng.choice([9, 10]))   # the fp32-input matrix-core variant's widths
    kind = rng.integers(0, 8)
    if big:
        n = int(rng.choice([9000, 40000, 150000, 400000]))
        kind = int(rng.integers(5, 8))
    c = gaussian_blobs(n, d, seed=int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.02, 0.08, 0.3])))
    if kind == 1:   # duplicates
        c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
    if kind == 2:   # large offset (cancellation stress)
        c += np.float32(rng.choice([10.0, 1000.0]))
    if kind == 3:   # tiny scale
        c *= np.float32(1e-3)
    if kind == 4:   # far from 1: the power-of-two scale of the fp16 operand images
        c *= np.float32(rng.choice([1e-12, 1e-6, 1e4, 1e8]))
    sig_loc = None
    if kind >= 5 and n > 1:   # clusters spread over the (col 0, col 1) plane: the components of the pruned sweeps (round 3)
        k = int(rng.choice([2, 3, 5, 12, 40, 90]))
        sig_loc = float(rng.choice([0.02, 0.08, 0.3]))
        spread = float(rng.choice([1.0, 4.0, 30.0, 300.0, 3000.0])) * sig_loc * np.sqrt(d)
        cen = np.zeros((k, d), dtype=np.float32)
        cen[:, :min(d, 2)] = rng.uniform(-spread, spread, size=(k, min(d, 2)))
        if kind == 7 and d > 2:   # the clusters differ in the other columns too
            cen[:, 2:] = rng.uniform(-spread, spread, size=(k, d - 2)) * 0.1
        lab = rng.integers(0, k, n)
        c = (cen[lab] + rng.normal(0.0, sig_loc, size=(n, d))).astype(np.float32)
        if kind == 6:           # a few far outliers
            c[rng.integers(0, n, max(1, n // 500))] += np.float32(50.0 * spread)
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    scale = float(np.sqrt(d)) * (sig_loc if sig_loc is not None else float(c.std(axis=0).mean() if n > 1 else 1.0))
    radii = [float(x) for x in (scale * rng.uniform(0.05, 1.5, size=int(rng.choice([1, 1, 2, 3, 4, 5, 8]))))]
    lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
    if rng.random() < 0.5: lo, hi = 0, n
    ref_p = dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, radii[:1], variant="direct")[0].contiguous())
    ref_n = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")
    for v in ("pruned", "mfma") + (("mfma32",) if d in (9, 10) else ()):
        p = dens.calculate_populations_partial(ct, radii, lo, hi, variant=v)
        q = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant=v)
        ok = bool((p == ref_p).all()) and all(bool((x.view(torch.int32) == y.view(torch.int32)).all()) for x, y in zip(q, ref_n))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} variant {v}: n={n} d={d} kind={kind} radii={radii} rows=[{lo},{hi}) pops_ok={bool((p == ref_p).all())}")
    # segments of a sharded run merge to the full result; the radius graph agrees with the populations
    if case % 3 == 0 and n > 1:
        G = int(rng.integers(2, 9))
        full_p = dens.calculate_populations_partial(ct, radii, variant="direct")
        full_n = dens.nearest_neighbors_partial(ct, fe, variant="direct")
        acc = torch.zeros_like(full_p)
        words = None
        for g in range(G):
            acc += dens.calculate_populations_segment(ct, radii, g, G)
            a, b, cc, dd = dens.nearest_neighbors_segment(ct, fe, g, G)
            w = torch.stack([(b.view(torch.int32).to(torch.int64) << 32) | (a.to(torch.int64) & 0xFFFFFFFF),
                             (dd.view(torch.int32).to(torch.int64) << 32) | (cc.to(torch.int64) & 0xFFFFFFFF)])
            words = w if words is None else torch.minimum(words, w)
        ok = bool((acc == full_p).all()) and bool(((words[0] & 0xFFFFFFFF).to(torch.int32) == full_n[0]).all()) \
            and bool(((words[0] >> 32).to(torch.int32) == full_n[1].view(torch.int32)).all()) \
            and bool(((words[1] & 0xFFFFFFFF).to(torch.int32) == full_n[2]).all()) \
            and bool(((words[1] >> 32).to(torch.int32) == full_n[3].view(torch.int32)).all())
        if not ok:
            bad += 1
            print(f"SEGMENT MISMATCH case {case}: n={n} d={d} G={G} kind={kind}")
        if n <= 40000:
            r2 = float(np.float32(radii[0]) * np.float32(radii[0]))
            try:
                pairs, pp = dens.radius_pairs(ct, r2)
            except Exception as e:
                print(f"RADIUS PAIRS ERROR case {case}: n={n} d={d} kind={kind} r2={r2} absmax={float(np.abs(c).max())}: {e}")
                bad += 1
                continue
            deg = torch.ones(n, dtype=torch.int64, device="cuda")
            if pairs.shape[0]:
                deg += torch.bincount(pairs.reshape(-1), minlength=n)
            if not (bool((deg == full_p[0].to(torch.int64)).all()) and bool((pp == full_p[0]).all())):
                bad += 1
                print(f"RADIUS PAIRS MISMATCH case {case}: n={n} d={d} r2={r2} pairs={pairs.shape[0]}")
print(f"{n_cases} cases, {bad} mismatches, {time.time()-t0:.1f}s")
