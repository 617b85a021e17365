"""Bounded, seeded fuzz of the matrix-core sweeps (-m gpu): random shapes, column counts 1..64, row ranges,
radii, duplicates, offsets and scales (the power-of-two scale of the fp16 operand images, the guard band
around cancellation) -- the pruned and the unpruned matrix-core variants against the direct VALU kernels
(exact by construction, themselves pinned to the oracle in test_gpu_parity.py), bit for bit; every third
case also merges the segments of a sharded run and checks the radius graph against the populations.
The open-ended version of the same loop is scratch/fuzz.py."""
import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu


def one_case(dens, rng, case, clustered=False):
    import torch
    n = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000, 3000, 9000, 40000],
                       p=[.03, .03, .05, .05, .05, .05, .1, .17, .2, .12, .08, .07]))
    d = int(rng.integers(1, 65)) if rng.random() < 0.3 else int(rng.integers(1, 33))
    kind = int(rng.integers(5, 8)) if clustered else int(rng.integers(0, 5))
    c = gaussian_blobs(n, d, seed=int(rng.integers(1, 1 << 30)), sigma=float(rng.choice([0.02, 0.08, 0.3])))
    if kind == 1:   # duplicates
        c[rng.integers(0, n, n // 3)] = c[rng.integers(0, n, n // 3)]
    if kind == 2:   # large offset (cancellation stress)
        c += np.float32(rng.choice([10.0, 1000.0]))
    if kind == 3:   # tiny scale
        c *= np.float32(1e-3)
    if kind == 4:   # far from 1
        c *= np.float32(rng.choice([1e-12, 1e-6, 1e4, 1e8]))
    sig_loc = None
    if kind >= 5 and n > 1:
        # clusters spread over the (col 0, col 1) plane, 1 ... 3000 cluster widths apart: the components of the pruned
        # sweeps -- one origin per cluster, cross-component pairs and neighbours, the one-component
        # fallbacks (90 clusters; clusters that touch)
        k = int(rng.choice([2, 3, 5, 12, 40, 90]))
        sig_loc = float(rng.choice([0.02, 0.08, 0.3]))
        spread = float(rng.choice([1.0, 4.0, 30.0, 300.0, 3000.0])) * sig_loc * np.sqrt(d)
        cen = np.zeros((k, d), dtype=np.float32)
        cen[:, :min(d, 2)] = rng.uniform(-spread, spread, size=(k, min(d, 2)))
        if kind == 7 and d > 2:   # the clusters differ in the other columns too
            cen[:, 2:] = rng.uniform(-spread, spread, size=(k, d - 2)) * 0.1
        c = (cen[rng.integers(0, k, n)] + rng.normal(0.0, sig_loc, size=(n, d))).astype(np.float32)
        if kind == 6:             # a few far outliers
            c[rng.integers(0, n, max(1, n // 500))] += np.float32(50.0 * spread)
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    scale = float(np.sqrt(d)) * (sig_loc if sig_loc is not None else float(c.std(axis=0).mean() if n > 1 else 1.0))
    radii = [float(x) for x in (scale * rng.uniform(0.05, 1.5, size=int(rng.integers(1, 4))))]
    lo = int(rng.integers(0, n))
    hi = int(rng.integers(lo, n + 1))
    if rng.random() < 0.5:
        lo, hi = 0, n
    what = f"case {case}: n={n} d={d} kind={kind} radii={radii} rows=[{lo},{hi})"
    ref_p = dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")
    fe = dens.calculate_free_energies(
        dens.calculate_populations_partial(ct, radii[:1], variant="direct")[0].contiguous())
    ref_n = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")
    for v in ("pruned", "mfma"):
        p = dens.calculate_populations_partial(ct, radii, lo, hi, variant=v)
        q = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant=v)
        assert bool((p == ref_p).all()), f"populations, {v}, {what}"
        for x, y in zip(q, ref_n):
            assert bool((x.view(torch.int32) == y.view(torch.int32)).all()), f"neighbours, {v}, {what}"
    if case % 3 == 0 and n > 1:
        G = int(rng.integers(2, 9))
        full_p = dens.calculate_populations_partial(ct, radii, variant="direct")
        full_n = dens.nearest_neighbors_partial(ct, fe, variant="direct")
        acc = torch.zeros_like(full_p)
        words = None
        for g in range(G):
            acc += dens.calculate_populations_segment(ct, radii, g, G)
            w = dens.pack_neighbors(*dens.nearest_neighbors_segment(ct, fe, g, G))
            words = w if words is None else torch.minimum(words, w)
        assert bool((acc == full_p).all()), f"segments (G={G}), {what}"
        for got, want in zip(dens.unpack_neighbors(words), full_n):
            assert bool((got.view(torch.int32) == want.view(torch.int32)).all()), f"segments (G={G}), {what}"
        r2 = float(np.float32(radii[0]) * np.float32(radii[0]))
        pairs, pp = dens.radius_pairs(ct, r2)
        deg = torch.ones(n, dtype=torch.int64, device="cuda")
        if pairs.shape[0]:
            deg += torch.bincount(pairs.reshape(-1), minlength=n)
        assert bool((deg == full_p[0].to(torch.int64)).all()) and bool((pp == full_p[0]).all()), f"radius pairs, {what}"


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_matrix_core_variants_against_direct(seed):
    import torch
    assert torch.cuda.is_available()
    from clustering_amd import density as dens
    rng = np.random.default_rng(seed)
    for case in range(40):
        one_case(dens, rng, case)


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_clustered_data_against_direct(seed):
    """the same loop on data made of 2 ... 90 clusters spread over the plane of the first two columns"""
    import torch
    assert torch.cuda.is_available()
    from clustering_amd import density as dens
    rng = np.random.default_rng(seed)
    for case in range(40):
        one_case(dens, rng, case, clustered=True)
