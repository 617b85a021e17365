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
    # ... and as the REFERENCE's own run printed them (mean 7233.1, max 65950)
    assert round(ph.sum() / n, 1) == 7233.1
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
    sigma2 = dens.compute_sigma2(nn[1])
    assert sigma2 == float(np.cumsum(nn_d2.astype(np.float64))[-1] / n)
    # the reference's own run of this workload (BASELINE.md:34,46): sigma2 = 0.00704766, the only
    # reference-derived value on the neighbour side; lumping radius sqrt(4 sigma2) = 0.1679
    assert abs(sigma2 - 0.00704766) < 5e-9
    assert abs(float(np.float32(np.sqrt(4.0 * sigma2))) - 0.1679) < 5e-5


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
        # the reference's own run of this workload: mean pops 2.8 / 719 / 9223
        means = pops.astype(np.float64).mean(axis=1)
        assert round(means[0], 1) == 2.8 and round(means[1]) == 719 and round(means[2]) == 9223, means
    for k in range(3):
        fe = dens.calculate_free_energies(torch.from_numpy(want[k].astype(np.int32)).cuda())
        assert (bits(fe.cpu().numpy()) == bits(fast_oracle.free_energies(want[k]))).all()
    assert (want[0] <= want[1]).all() and (want[1] <= want[2]).all()          # monotone in the radius


C5_RADII = [0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 0.65]


def test_c5_one_segment_of_eight_all_radii_and_neighbours(dens, fast_oracle):
    """C5 = 5M x 30, 8 radii, on 8 GPUs (BASELINE.json configs[4]): what the ranks compute.  Populations: the eight
    radii go through ONE symmetric sweep per rank (pop_msym_kernel: every unordered pair of query groups once, both
    frames credited), so a segment's counts are PARTIAL counts of all rows -- the eight segments are run one after the
    other and SUMMED (the all-reduce of a real run) and the sums are compared with the oracle on a row block (each row
    against all 5M reference frames) and with the row-block call of the reference's partition.  Neighbours: segment 3
    of 8, nn / nn_hd against the oracle on > 500 of its rows, and size-independent properties on every row."""
    import torch
    n, d, G, seg_id = 5_000_000, 30, 8, 3
    c = gaussian_blobs(n, d)
    ct = torch.from_numpy(c).cuda()
    total = None
    for g in range(G):
        part = dens.calculate_populations_segment(ct, C5_RADII, g, G)
        if g == seg_id:
            seg_part = part.cpu().numpy().astype(np.uint32)
        total = part if total is None else total + part
    seg = total.cpu().numpy().astype(np.uint32)       # populations of ALL rows, all eight radii
    del total, part
    assert (seg_part <= seg).all() and 0.05 * seg[7].astype(np.int64).sum() < seg_part[7].astype(np.int64).sum() < 0.25 * seg[7].astype(np.int64).sum()
    assert (seg[0] >= 1).all()
    for k in range(1, len(C5_RADII)):                 # monotone in the radius
        assert (seg[k] >= seg[k - 1]).all()
    for k in range(len(C5_RADII)):                    # every pair is counted at both ends
        assert (seg[k].astype(np.int64).sum() - n) % 2 == 0
    rows = np.arange(n)
    lo, width = n // 2 + 1234, 600                    # a row block: every row against all 5M references
    want = fast_oracle.populations(c, C5_RADII, lo, lo + width)
    assert (seg[:, lo:lo + width].astype(np.uint64) == want[:, lo:lo + width]).all()
    block = dens.calculate_populations_partial(ct, C5_RADII, lo, lo + 700).cpu().numpy().astype(np.uint32)
    assert (block[:, lo:lo + 600].astype(np.uint64) == want[:, lo:lo + 600]).all()
    assert (block[:, lo:lo + 700] == seg[:, lo:lo + 700]).all()
    width = 4400                                      # (the neighbour check below: a block that holds > 500 rows of the segment)
    # free energies of ALL frames at r = 0.5 (a real run all-reduces the eight segments; here one full sweep)
    full = dens.calculate_populations_partial(ct, [0.5])
    fh = full[0].cpu().numpy().astype(np.uint32)
    assert (fh[rows] == seg[4][rows]).all() and (fh.astype(np.int64).sum() - n) % 2 == 0
    fe = dens.calculate_free_energies(full[0].contiguous())
    fe_h = fe.cpu().numpy()
    assert (bits(fe_h) == bits(fast_oracle.free_energies(fh.astype(np.uint64)))).all()
    nn = dens.nearest_neighbors_segment(ct, fe, seg_id, G)
    nn_idx, nn_d2, hd_idx, hd_d2 = [t.cpu().numpy() for t in nn]
    nn_idx = nn_idx.astype(np.uint32).astype(np.int64)
    hd_idx = hd_idx.astype(np.uint32).astype(np.int64)
    # the neighbour sweep cuts its OWN spatial order (cells of columns 0/1, then free energy) into segments:
    # segment 3 of the neighbour sweep holds other rows than segment 3 of the population sweep (any
    # partition serves a sharded run, as long as the partials of each sweep merge)
    fmax = np.finfo(np.float32).max
    nrows = np.nonzero(nn_idx <= n)[0]
    assert abs(len(nrows) - n / G) < 0.02 * n
    other = np.ones(n, dtype=bool)
    other[nrows] = False
    assert (nn_idx[other] == n + 1).all() and (nn_d2[other] == fmax).all()
    assert (hd_idx[other] == n + 1).all() and (hd_d2[other] == fmax).all()
    mine_n = nrows[(nrows >= lo) & (nrows < lo + width)]
    assert len(mine_n) >= 500
    exp = fast_oracle.nearest_neighbors(c, fe_h, lo, lo + width)
    assert (nn_idx[mine_n] == exp[0][mine_n].astype(np.int64)).all() and (hd_idx[mine_n] == exp[2][mine_n].astype(np.int64)).all()
    assert (bits(nn_d2[mine_n]) == bits(exp[1][mine_n])).all() and (bits(hd_d2[mine_n]) == bits(exp[3][mine_n])).all()
    # properties on all rows of the segment
    assert (nn_idx[nrows] != nrows).all() and (nn_idx[nrows] < n).all() and (nn_d2[nrows] <= hd_d2[nrows]).all()
    has_hd = nrows[hd_idx[nrows] < n]
    assert (fe_h[hd_idx[has_hd]] < fe_h[has_hd]).all()
    no_hd = nrows[hd_idx[nrows] > n]
    assert (fe_h[no_hd] == fe_h.min()).all() and (hd_d2[no_hd] == fmax).all()
    rng = np.random.default_rng(5)
    sample = rng.choice(nrows, 3000, replace=False)
    assert (bits(canonical_d2_rows(fast_oracle, c, sample, nn_idx[sample])) == bits(nn_d2[sample])).all()
    hs = sample[hd_idx[sample] < n]
    assert (bits(canonical_d2_rows(fast_oracle, c, hs, hd_idx[hs])) == bits(hd_d2[hs])).all()
    # a neighbour can be no farther than the population radius allows: pop(r) > 1  <=>  nn_d2 < r^2
    both = np.intersect1d(rows, nrows)
    assert len(both) > 0.005 * n
    for k, r in enumerate(C5_RADII):
        r2 = np.float32(r) * np.float32(r)
        assert ((seg[k][both] > 1) == (nn_d2[both] < r2)).all()


def test_screening_forest_at_30_dims(dens, fast_oracle, tmp_path):
    """-T at D = 30 (C5's screened lumping, density_clustering.cpp:773-817) on 240 000 frames: the GPU's
    bottleneck spanning forest + the reference's name bookkeeping must reproduce, threshold by threshold,
    the quadratic restatement (oracle/screening_oracle.cpp) run on the frames below each threshold."""
    import os
    import subprocess
    import torch
    from oracle.oracle import ScreeningOracle
    from clustering_amd import capi
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "bin", "clustering")
    n, d, r = 240_000, 30, 0.5
    c = gaussian_blobs(n, d, seed=77)
    np.save(tmp_path / "c.npy", c)
    ct = torch.from_numpy(c).cuda()
    pops = dens.calculate_populations_partial(ct, [r])
    fe = dens.calculate_free_energies(pops[0].contiguous())
    nn = dens.nearest_neighbors_partial(ct, fe)
    fe_h, nn_d2 = fe.cpu().numpy(), nn[1].cpu().numpy()
    # thresholds: three levels that put 1 % .. 6 % of the frames below them (the restatement is quadratic in those)
    q = np.quantile(fe_h, [0.01, 0.06])
    t_from = float(np.round(q[0], 2))
    step = float(np.round((q[1] - q[0]) / 2.0, 2))
    assert step > 0.0
    t_to = t_from + 2.0 * step
    run = subprocess.run([cli, "density", "-f", str(tmp_path / "c.npy"), "-r", str(r), "-T", "%.2f" % t_from,
                          "%.2f" % step, "%.2f" % t_to, "-o", str(tmp_path / "clust"), "-v"],
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "span the graph" in run.stdout + run.stderr
    so = ScreeningOracle()
    clustering, n_files = None, 0
    t, st, tt = np.float32("%.2f" % t_from), np.float32("%.2f" % step), np.float32("%.2f" % t_to)
    while t < tt - st / np.float32(10.0) + st and not (tt + st / np.float32(10.0) + st < t):
        clustering = so.screening(fe_h, nn_d2, t, c, clustering)
        with open(str(tmp_path / "clust") + ".%0.2f" % t) as f:
            got = np.array([int(x) for x in f if x.strip() and not x.startswith("#")], dtype=np.uint64)
        assert (got == clustering).all(), f"threshold {t}"
        assert (got != 0).sum() == (fe_h < t).sum()
        n_files += 1
        t = np.float32(t + st)
    assert n_files == 3 and (clustering != 0).sum() > 0.04 * n


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
