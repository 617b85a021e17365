"""GPU tests at BASELINE.json's full sizes (run with -m gpu on an MI355X).

The quadratic oracle cannot sweep 10^12 pairs, so at these sizes the HIP path is checked
  - against the oracle on row ranges (the oracle takes (i_from, i_to): a few thousand query rows
    against ALL reference frames, canonical arithmetic, bit-exact),
  - against itself: the pruned matrix-core sweep, the unpruned one and the direct VALU kernels
    (exact by construction) must agree bit for bit on every row,
  - through size-independent properties: every pair is counted at both ends (sum of pop-1 is even),
    a neighbour's distance re-computed in the canonical order equals the stored one, nn is never
    farther than the neighbour of lower free energy, that neighbour HAS a lower free energy,
    "nearest" is mutual-consistent (my neighbour's neighbour is at most as far), and the check
    values of the normative generator.
"""
import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dens():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from clustering_amd import density
    return density


@pytest.fixture(scope="module")
def fast_oracle():
    """canonical arithmetic, all host threads (the row-range checks sweep 10^9..10^10 pairs)"""
    from oracle.oracle import Oracle, build
    build()
    return Oracle()


def canonical_d2_rows(oracle, c, i, j):
    """canonical d2 of the row pairs (i[k], j[k]) by the oracle's single-pair entry point"""
    return np.array([oracle.dist2(c[a], c[b]) for a, b in zip(i, j)], dtype=np.float32)


def test_c3_full_path(dens, fast_oracle):
    """C3 = 1M x 10, r = 0.2, pop + FE + nn on one GPU (the bench workload)."""
    import torch
    n, d, r = 1_000_000, 10, 0.2
    c = gaussian_blobs(n, d)
    ct = torch.from_numpy(c).cuda()
    pops = dens.calculate_populations_partial(ct, [r])                      # default: pruned
    p_mfma = dens.calculate_populations_partial(ct, [r], variant="mfma")
    p_direct = dens.calculate_populations_partial(ct, [r], variant="direct")
    assert bool((pops == p_mfma).all()) and bool((pops == p_direct).all())
    ph = pops[0].cpu().numpy().astype(np.int64)
    # check values of the normative generator (seed 20240), as every variant has produced them
    assert ph.sum() == 7233139928 and ph.max() == 65950
    assert (ph.sum() - n) % 2 == 0 and ph.min() >= 1
    rng = np.random.default_rng(3)
    starts = [0, int(rng.integers(1000, n - 3000)), n - 1500]
    for s in starts:
        want = fast_oracle.populations(c, [r], s, s + 1500)[0][s:s + 1500]
        assert (ph[s:s + 1500] == want.astype(np.int64)).all(), f"pops rows {s}.."
    fe = dens.calculate_free_energies(pops[0].contiguous())
    fe_h = fe.cpu().numpy()
    fe_want = fast_oracle.free_energies(ph.astype(np.uint64))
    assert (bits(fe_h) == bits(fe_want)).all()
    nn = dens.nearest_neighbors_partial(ct, fe)
    nn_direct = dens.nearest_neighbors_partial(ct, fe, variant="direct")
    for a, b in zip(nn, nn_direct):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all())
    nn_idx, nn_d2, hd_idx, hd_d2 = [t.cpu().numpy() for t in nn]
    nn_idx = nn_idx.astype(np.uint32).astype(np.int64)
    hd_idx = hd_idx.astype(np.uint32).astype(np.int64)
    for s in starts[:2]:
        exp = fast_oracle.nearest_neighbors(c, fe_h, s, s + 600)
        sl = slice(s, s + 600)
        assert (nn_idx[sl] == exp[0][sl].astype(np.int64)).all() and (hd_idx[sl] == exp[2][sl].astype(np.int64)).all()
        assert (bits(nn_d2[sl]) == bits(exp[1][sl])).all() and (bits(hd_d2[sl]) == bits(exp[3][sl])).all()
    # properties on all rows
    assert (nn_idx != np.arange(n)).all() and (nn_idx < n).all()
    assert (nn_d2 <= hd_d2).all()
    has_hd = hd_idx < n
    assert (~has_hd).sum() >= 1                       # the frames at the free-energy minimum have none
    assert (fe_h[hd_idx[has_hd]] < fe_h[has_hd]).all()
    assert (fe_h[~has_hd] == fe_h.min()).all() and (hd_d2[~has_hd] == np.finfo(np.float32).max).all()
    assert (nn_d2[nn_idx] <= nn_d2).all()             # my neighbour has a neighbour at least as close
    sample = rng.integers(0, n, 4000)
    assert (bits(canonical_d2_rows(fast_oracle, c, sample, nn_idx[sample])) == bits(nn_d2[sample])).all()
    hs = sample[has_hd[sample]]
    assert (bits(canonical_d2_rows(fast_oracle, c, hs, hd_idx[hs])) == bits(hd_d2[hs])).all()
    # sigma2: double sum in frame order (density_clustering.cpp:334-343); numpy's cumsum adds sequentially
    assert dens.compute_sigma2(nn[1]) == float(np.cumsum(nn_d2.astype(np.float64))[-1] / n)


def test_c2_three_radii_against_the_oracle(dens, fast_oracle):
    """C2 = 100k x 10, radii {0.1, 0.2, 0.3} in one call, pop + FE: the whole oracle sweep (10^10 pairs)."""
    import torch
    n, d, radii = 100_000, 10, [0.1, 0.2, 0.3]
    c = gaussian_blobs(n, d)
    ct = torch.from_numpy(c).cuda()
    want = fast_oracle.populations(c, radii)
    for variant in ("pruned", "mfma", "direct"):
        pops = dens.calculate_populations_partial(ct, radii, variant=variant).cpu().numpy()
        assert (pops.astype(np.uint32).astype(np.uint64) == want).all(), variant
    for k in range(3):
        fe = dens.calculate_free_energies(torch.from_numpy(want[k].astype(np.int32)).cuda())
        assert (bits(fe.cpu().numpy()) == bits(fast_oracle.free_energies(want[k]))).all()
    assert (want[0] <= want[1]).all() and (want[1] <= want[2]).all()          # monotone in the radius


def test_c5_shape_one_segment_of_eight(dens, fast_oracle):
    """C5 = 5M x 30 on 8 GPUs: what ONE rank computes (segment 3 of 8) for two of the eight radii, against
    the oracle on rows of that segment, and against the row-block call of the reference's partition."""
    import torch
    n, d, radii = 5_000_000, 30, [0.45, 0.65]
    c = gaussian_blobs(n, d)
    ct = torch.from_numpy(c).cuda()
    seg = dens.calculate_populations_segment(ct, radii, 3, 8).cpu().numpy().astype(np.uint32)
    rows = np.nonzero(seg[0])[0]                      # populations are >= 1 on the rows of the segment
    assert abs(len(rows) - n / 8) < 0.02 * n and (seg[1][rows] >= seg[0][rows]).all()
    assert (np.nonzero(seg[1])[0] == rows).all()
    lo = int(rows[len(rows) // 2])
    block = dens.calculate_populations_partial(ct, radii, lo, lo + 700).cpu().numpy().astype(np.uint32)
    want = fast_oracle.populations(c, radii, lo, lo + 700)
    assert (block[:, lo:lo + 700].astype(np.uint64) == want[:, lo:lo + 700]).all()
    mine = rows[(rows >= lo) & (rows < lo + 700)]
    assert len(mine) > 20 and (seg[:, mine] == block[:, mine]).all()


def test_c4_eight_segments_merge_to_the_single_device_result(dens):
    """C4 = C3 sharded over 8 GPUs: the eight segments a rank each would compute, merged the way
    clustering_amd.distributed merges them (sum of the populations, minimum of the packed neighbour words),
    equal the single-device result on every row."""
    import torch
    n, d, r, G = 1_000_000, 10, 0.2, 8
    ct = torch.from_numpy(gaussian_blobs(n, d)).cuda()
    full_p = dens.calculate_populations_partial(ct, [r])
    fe = dens.calculate_free_energies(full_p[0].contiguous())
    full_n = dens.nearest_neighbors_partial(ct, fe)
    acc_p = torch.zeros_like(full_p)
    words = None
    for g in range(G):
        acc_p += dens.calculate_populations_segment(ct, [r], g, G)
        w = dens.pack_neighbors(*dens.nearest_neighbors_segment(ct, fe, g, G))
        words = w if words is None else torch.minimum(words, w)
    assert bool((acc_p == full_p).all())
    for got, want in zip(dens.unpack_neighbors(words), full_n):
        assert bool((got.view(torch.int32) == want.view(torch.int32)).all())
