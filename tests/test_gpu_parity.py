"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs -- bit-exact populations, neighbour indices and
d2 bits; free energies bit-exact as well (same host libm), asserted within the 1e-5 relative
tolerance BASELINE.json states plus bitwise as a stronger check."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kat_cases.json")))["cases"]
FLT_MAX = np.finfo(np.float32).max
VARIANTS = ["direct", "mfma", "pruned"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dens():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from clustering_amd import density
    return density


def need(variant, n_cols):
    if not _supported(variant, n_cols):
        pytest.skip(f"{variant} variant does not support n_cols={n_cols}")


def _supported(variant, n_cols):
    if variant == "direct":
        return True
    from clustering_amd import capi
    return capi.lib.dc_hip_workspace_bytes(64, n_cols, 1) > 0


def run_gpu(dens, c, radii, fe_from, variant, i_from=0, i_to=None):
    import torch
    ct = torch.from_numpy(c).cuda()
    pops = dens.calculate_populations_partial(ct, radii, i_from, i_to, variant=variant)
    fe = dens.calculate_free_energies(pops[fe_from].contiguous())
    return pops.cpu().numpy().astype(np.uint32), fe, ct


def check_full(dens, oracle, c, radii, variant, fe_from=0):
    import torch
    pops, fe, ct = run_gpu(dens, c, radii, fe_from, variant)
    want = oracle.populations(c, radii)
    assert (pops.astype(np.uint64) == want).all(), f"pops mismatch ({variant})"
    fe_want = oracle.free_energies(want[fe_from])
    fe_got = fe.cpu().numpy()
    np.testing.assert_allclose(fe_got, fe_want, rtol=1e-5, atol=1e-7)   # north_star tolerance
    assert (bits(fe_got) == bits(fe_want)).all()                        # and in fact bit-exact
    nn = dens.nearest_neighbors_partial(ct, fe, variant=variant)
    exp = oracle.nearest_neighbors(c, fe_want)
    got = [t.cpu().numpy() for t in nn]
    assert (got[0].astype(np.uint32).astype(np.uint64) == exp[0]).all(), f"nn idx ({variant})"
    assert (got[2].astype(np.uint32).astype(np.uint64) == exp[2]).all(), f"hd idx ({variant})"
    assert (bits(got[1]) == bits(exp[1])).all(), f"nn d2 bits ({variant})"
    assert (bits(got[3]) == bits(exp[3])).all(), f"hd d2 bits ({variant})"
    s2 = dens.compute_sigma2(nn[1])
    assert s2 == oracle.sigma2(exp[1])


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", KATS, ids=[c["name"] for c in KATS])
def test_kat(dens, case, variant):
    import torch
    c = np.array(case["coords"], dtype=np.float32)
    if not _supported(variant, c.shape[1]):
        pytest.skip("variant does not support this n_cols")
    radii = case["radii"]
    sel = radii.index(case["fe_from_radius"])
    pops, fe, ct = run_gpu(dens, c, radii, sel, variant)
    assert pops.tolist() == case["pops"]
    nn = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fe, variant=variant)]
    assert nn[0].tolist() == case["nn_idx"] and nn[2].tolist() == case["hd_idx"]
    assert (bits(nn[1]) == bits(np.array(case["nn_d2"], np.float32))).all()
    assert (bits(nn[3]) == bits(np.array(case["hd_d2"], np.float32))).all()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 7, 10, 12, 13, 16, 24, 25, 30, 32])
def test_parity_templated_dims(dens, oracle, D, variant):
    if not _supported(variant, D):
        pytest.skip("variant does not support this n_cols")
    c = gaussian_blobs(3000, D, seed=1000 + D)
    radii = [0.2] if D <= 10 else [0.08 * np.sqrt(2.0 * D)]
    check_full(dens, oracle, c, radii, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,D", [(3000, 10), (4111, 9), (20000, 10), (33, 10)])
def test_fp32_mfma_variant_against_the_oracle(dens, oracle, n_rows, D):
    """DC_VARIANT_MFMA32 (dc_mfma32.hpp): the literal fp32-input MFMA instance (v_mfma_f32_32x32x2_f32, n_cols 9..10,
    every pair) that BASELINE.json's "fraction of the fp32 MFMA roofline" is quoted on -- populations (two radii, a row
    range), free energies and nn / nn_hd with their d2 bits against the oracle, duplicates included; other column
    counts are refused."""
    import torch
    c = gaussian_blobs(n_rows, D, seed=4000 + n_rows)
    if n_rows > 100:
        rng = np.random.default_rng(n_rows)
        c[rng.integers(0, n_rows, n_rows // 9)] = c[rng.integers(0, n_rows, n_rows // 9)]   # ties: lowest index wins
    check_full(dens, oracle, c, [0.2, 0.11], "mfma32")
    ct = torch.from_numpy(c).cuda()
    lo, hi = n_rows // 3, n_rows // 3 + max(1, n_rows // 2)
    got = dens.calculate_populations_partial(ct, [0.2], lo, hi, variant="mfma32").cpu().numpy().astype(np.uint32)
    want = oracle.populations(c, [0.2], lo, hi)
    assert (got.astype(np.uint64) == want).all()
    with pytest.raises(RuntimeError):
        dens.calculate_populations_partial(torch.from_numpy(gaussian_blobs(500, 12, seed=1)).cuda(), [0.3], variant="mfma32")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("D", [33, 40, 48, 64, 65, 100])
def test_parity_generic_dims(dens, oracle, D, variant):
    if not _supported(variant, D):
        pytest.skip("variant does not support this n_cols")
    c = gaussian_blobs(1500, D, seed=2000 + D)
    check_full(dens, oracle, c, [0.08 * np.sqrt(2.0 * D), 0.1 * np.sqrt(2.0 * D)], variant, fe_from=1)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_rows,D", [(64, 37), (100, 61), (257, 64), (33, 48)])
def test_parity_wide_rows_small_sets(dens, oracle, n_rows, D, variant):
    """33 .. 64 columns on a handful of frames (the header of the workspace holds per-column sums and
    means: sized for 64 columns; a fuzz run caught it while it was still sized for 32)"""
    need(variant, D)
    c = gaussian_blobs(n_rows, D, seed=3000 + D)
    check_full(dens, oracle, c, [0.08 * np.sqrt(2.0 * D), 0.11 * np.sqrt(2.0 * D)], variant)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_radii", [1, 2, 3, 4, 5, 8, 9, 17])
def test_parity_multi_radius(dens, oracle, n_radii, variant):
    need(variant, 10)
    c = gaussian_blobs(2500, 10, seed=31)
    radii = list(np.linspace(0.35, 0.05, n_radii).astype(np.float32))  # descending, as -R sorts them
    if n_radii >= 3:
        radii[0], radii[2] = radii[2], radii[0]                        # and unsorted input order
    import torch
    pops = dens.calculate_populations_partial(torch.from_numpy(c).cuda(), radii, variant=variant)
    assert (pops.cpu().numpy().astype(np.uint32).astype(np.uint64) == oracle.populations(c, radii)).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_parity_c1_config(dens, oracle, variant):
    """BASELINE.json configs[0]: 10k x 5, single radius 0.1 -- full path, plus the reference-run
    statistics of BASELINE.md (mean pop 73.6, max 314)."""
    need(variant, 5)
    c = gaussian_blobs(10000, 5)
    check_full(dens, oracle, c, [0.1], variant)
    pops = oracle.populations(c, [0.1])[0]
    assert pops.max() == 314 and round(float(pops.mean()), 1) == 73.6


@pytest.mark.parametrize("variant", VARIANTS)
def test_ragged_sizes_and_row_ranges(dens, oracle, variant):
    import torch
    need(variant, 10)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1023, 1025):
        c = gaussian_blobs(n, 10, seed=500 + n)
        want = oracle.populations(c, [0.2, 0.3])
        ct = torch.from_numpy(c).cuda()
        got = dens.calculate_populations_partial(ct, [0.2, 0.3], variant=variant).cpu().numpy()
        assert (got.astype(np.uint32).astype(np.uint64) == want).all(), n
        fe = oracle.free_energies(want[0])
        nn = dens.nearest_neighbors_partial(ct, torch.from_numpy(fe).cuda(), variant=variant)
        exp = oracle.nearest_neighbors(c, fe)
        assert (nn[0].cpu().numpy().astype(np.uint32).astype(np.uint64) == exp[0]).all(), n
        assert (nn[2].cpu().numpy().astype(np.uint32).astype(np.uint64) == exp[2]).all(), n
        assert (bits(nn[1].cpu().numpy()) == bits(exp[1])).all(), n
        assert (bits(nn[3].cpu().numpy()) == bits(exp[3])).all(), n
    # partial row ranges: zero / "none" outside, partials merge by sum / by ownership
    c = gaussian_blobs(1500, 10, seed=77)
    ct = torch.from_numpy(c).cuda()
    full = oracle.populations(c, [0.2])
    fe = oracle.free_energies(full[0])
    fet = torch.from_numpy(fe).cuda()
    exp = oracle.nearest_neighbors(c, fe)
    acc = np.zeros_like(full)
    for lo, hi in ((0, 0), (0, 500), (500, 501), (501, 1500)):
        p = dens.calculate_populations_partial(ct, [0.2], lo, hi, variant=variant).cpu().numpy()
        p = p.astype(np.uint32).astype(np.uint64)
        assert (p[:, :lo] == 0).all() and (p[:, hi:] == 0).all()
        acc += p
        nn = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fet, lo, hi, variant=variant)]
        idx = nn[0].astype(np.uint32).astype(np.uint64)
        assert (idx[lo:hi] == exp[0][lo:hi]).all()
        assert (idx[:lo] == 1501).all() and (idx[hi:] == 1501).all()
        assert (nn[1][:lo] == FLT_MAX).all() and (nn[3][hi:] == FLT_MAX).all()
        assert (nn[2].astype(np.uint32).astype(np.uint64)[lo:hi] == exp[2][lo:hi]).all()
    assert (acc == full).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_duplicates_and_offset_data(dens, oracle, variant):
    """duplicated frames (d2 == 0 neighbours), exact ties, and data far from the origin (large
    |x|^2 stresses the Gram-form guard band of the MFMA variant)."""
    need(variant, 10)
    need(variant, 6)
    rng = np.random.default_rng(5)
    base = gaussian_blobs(700, 10, seed=9)
    c = np.concatenate([base, base[:300], base[100:150]]).astype(np.float32)   # duplicates
    c = c[rng.permutation(c.shape[0])]
    check_full(dens, oracle, c, [0.2, 0.25], variant)
    lattice = rng.integers(0, 4, (1200, 6)).astype(np.float32) * 0.25        # massive exact ties
    check_full(dens, oracle, lattice, [0.25, 0.5, 0.3535534], variant)
    shifted = (gaussian_blobs(1500, 10, seed=10) + np.float32(37.5)).astype(np.float32)
    check_full(dens, oracle, shifted, [0.2], variant)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("scale,D", [(1e-12, 10), (1e-6, 3), (1e-3, 10), (1e4, 10), (1e8, 30), (3e17, 5)])
def test_data_far_from_unit_scale(dens, oracle, scale, D, variant):
    """The fp16 operand images of the matrix-core sweeps carry a per-data-set power-of-two scale
    (scale_of in dc_mfma_kernels.hpp): results must not depend on the magnitude of the coordinates.
    Radii scale with the data, so the populations are those of the unit-scale set."""
    need(variant, D)
    base = gaussian_blobs(1300, D, seed=50 + D)
    c = (base * np.float32(scale)).astype(np.float32)
    r = float(np.sqrt(D) * 0.08 * 1.1)
    check_full(dens, oracle, c, [r * scale, 0.6 * r * scale], variant)
    # and a radius far beyond the data extent / far below the closest pair
    check_full(dens, oracle, c, [1e6 * scale, 1e-9 * scale], variant)


@pytest.mark.parametrize("variant", VARIANTS)
def test_non_finite_rows(dens, oracle, variant):
    """rows with inf / NaN coordinates: never inside any radius, never anybody's neighbour (every
    comparison with NaN/inf d2 is false, as in the reference); the MFMA variant hands such inputs
    to the exact kernels through its on-device gate."""
    need(variant, 10)
    c = gaussian_blobs(900, 10, seed=21)
    c[17, 3] = np.inf
    c[400, 0] = np.nan
    c[401, 9] = -np.inf
    check_full(dens, oracle, c, [0.2, 0.3], variant)
    huge = (gaussian_blobs(600, 10, seed=22) * np.float32(1e19)).astype(np.float32)   # |x|^2 overflows
    check_full(dens, oracle, huge, [2e18], variant)


def test_free_energies_device_log_with_host_referee(oracle):
    """dc_hip_free_energies_dev: device double log + host libm for the rows near a float rounding boundary
    (default) and the host table path (DC_FE_HOST_TABLE=1, in a child process: the switch is read once)
    both give the oracle's bits -- on every population value 0..max for several maxima, so that every
    boundary case of q = pop * (1/max) occurs."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from clustering_amd import density as dens
from oracle.oracle import Oracle, build
build(); o = Oracle()
bad = 0
for mx in (1, 2, 3, 1000, 65950, 1 << 20, (1 << 24) + 1, 3000001):
    step = max(1, mx // 700000)
    p = np.arange(0, mx + 1, step, dtype=np.int64)
    p[-1] = mx
    p = p.astype(np.int32)
    fe = dens.calculate_free_energies(torch.from_numpy(p).cuda()).cpu().numpy()
    want = o.free_energies(p.astype(np.uint64))
    bad += int((fe.view(np.uint32) != want.view(np.uint32)).sum())
print("mismatches", bad)
'''
    # default margin; the host table; a margin that sends thousands of rows to the referee; one that
    # overflows the referee's list (falls back to the table)
    for table, tol in (("0", "0"), ("1", "0"), ("0", "2e-11"), ("0", "1e-6")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(HERE),
                           env=dict(os.environ, DC_FE_HOST_TABLE=table, DC_FE_REFEREE_TOL=tol))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "mismatches 0" in r.stdout, (table, tol, r.stdout[-500:])


def test_host_pointer_entry_points(oracle):
    """dc_hip_populations / dc_hip_nearest_neighbors / dc_hip_density_all with HOST pointers."""
    from clustering_amd import capi
    c = gaussian_blobs(2000, 10, seed=3)
    n, d = c.shape
    radii = np.array([0.2, 0.1], dtype=np.float32)
    pops = np.zeros((2, n), dtype=np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    capi.check(capi.lib.dc_hip_populations(vp(c), n, d, vp(radii), 2, 100, 1900, 0, vp(pops)))
    want = oracle.populations(c, radii, 100, 1900)
    assert (pops.astype(np.uint64) == want).all()
    full = oracle.populations(c, radii)
    fe = oracle.free_energies(full[0])
    out = [np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros(n, np.float32)]
    capi.check(capi.lib.dc_hip_nearest_neighbors(vp(c), n, d, vp(fe), 0, n, 0, *[vp(a) for a in out]))
    exp = oracle.nearest_neighbors(c, fe)
    assert (out[0].astype(np.uint64) == exp[0]).all() and (out[2].astype(np.uint64) == exp[2]).all()
    assert (bits(out[1]) == bits(exp[1])).all() and (bits(out[3]) == bits(exp[3])).all()
    # whole path, one device
    pops2 = np.zeros((2, n), dtype=np.uint32)
    fe2 = np.zeros(n, np.float32)
    out2 = [np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros(n, np.float32)]
    capi.check(capi.lib.dc_hip_density_all(vp(c), n, d, vp(radii), 2, 0, 1, vp(pops2), vp(fe2),
                                           *[vp(a) for a in out2]))
    assert (pops2.astype(np.uint64) == full).all()
    assert (bits(fe2) == bits(fe)).all()
    assert (out2[0].astype(np.uint64) == exp[0]).all() and (bits(out2[3]) == bits(exp[3])).all()


def test_errors_are_codes_not_exits():
    from clustering_amd import capi
    c = np.zeros((4, 3), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    pops = np.zeros(4, np.uint32)
    r = np.array([1.0], np.float32)
    rc = capi.lib.dc_hip_populations(vp(c), 4, 3, vp(r), 1, 3, 2, 0, vp(pops))   # i_from > i_to
    assert rc == -1 and b"row range" in capi.lib.dc_hip_last_error()
    rc = capi.lib.dc_hip_populations(vp(c), 4, 3, vp(r), 1, 0, 4, 99, vp(pops))  # bad device
    assert rc == -1
    rc = capi.lib.dc_hip_populations(vp(c), 4, 0, vp(r), 1, 0, 4, 0, vp(pops))   # n_cols == 0
    assert rc == -1


@pytest.mark.gpu
def test_mfma_accumulate_model_and_gram_band():
    """The guard band of the matrix-core sweeps assumes (1) how v_mfma_f32_32x32x16_bf16 rounds its
    17-term sums and (2) that the bf16x3 Gram chain stays inside the MFMA part of the band; both are
    checked on the device by a small HIP program built with the library (tests/cpp/test_mfma_model.hip)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clustering_amd", "bin",
                       "test_mfma_model")
    assert os.path.exists(exe), "build the library first (make -C clustering_amd/csrc)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


@pytest.mark.gpu
def test_library_radix_sort_is_a_stable_sort():
    """The frame orders of the pruned sweeps come from the library's own radix sort (dc_sort.hip, round 5); the ranks of a
    sharded run must derive the SAME order, so it has to be a stable sort -- checked against std::stable_sort by a small
    HIP program built from the library's source (tests/cpp/test_sort.hip): sizes around the tile and wave boundaries,
    1 .. 32 key bits, heavy ties, all-equal keys."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clustering_amd", "bin", "test_sort")
    assert os.path.exists(exe), "build the library first (make -C clustering_amd/csrc)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "cases OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _brute_pairs(c, r2):
    """All unordered pairs with canonical d2 < r2 (numpy emulation of the canonical order)."""
    from refmath import d2_matrix
    d2 = d2_matrix(c)
    ii, jj = np.nonzero(np.tril(d2 < np.float32(r2), k=-1))   # j < i
    return {(int(j), int(i)) for i, j in zip(ii, jj)}


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_cols,r2", [(700, 3, 0.02), (3000, 10, 0.045), (1500, 30, 0.25), (257, 2, 1e-4)])
def test_radius_pairs_match_brute_force(dens, n_rows, n_cols, r2):
    """dc_hip_radius_pairs_dev lists exactly the pairs the reference's high_density_neighborhood would
    find frame by frame (strict <, canonical d2), each pair once; pops agree with the population sweep."""
    import torch
    from clustering_amd.synth import gaussian_blobs
    c = gaussian_blobs(n_rows, n_cols, seed=7 + n_cols)
    ct = torch.from_numpy(c).cuda()
    pairs, pops = dens.radius_pairs(ct, r2)
    got = [(int(min(a, b)), int(max(a, b))) for a, b in pairs.cpu().numpy()]
    assert len(got) == len(set(got)), "a pair was listed twice"
    want = _brute_pairs(c, r2)
    assert set(got) == want
    deg = np.ones(n_rows, dtype=np.int64)
    for a, b in want:
        deg[a] += 1
        deg[b] += 1
    assert (pops.cpu().numpy().astype(np.int64) == deg).all()


def _components(n, pairs):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in pairs:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return np.array([find(i) for i in range(n)])


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_cols,r2,n_comp", [(700, 3, 0.02, 700), (3000, 10, 0.045, 40), (1500, 30, 0.25, 5),
                                                     (257, 2, 1e-4, 257), (2000, 10, 0.06, 1)])
def test_min_edge_round_matches_brute_force(dens, n_rows, n_cols, r2, n_comp):
    """dc_hip_radius_min_edge_dev: for every component the lightest pair (max rank, min rank) of the
    radius graph that leaves it -- against the brute-force pair set and arbitrary component labels."""
    import torch
    from clustering_amd.synth import gaussian_blobs
    c = gaussian_blobs(n_rows, n_cols, seed=11 + n_cols)
    rng = np.random.default_rng(n_rows)
    rank = rng.permutation(n_rows).astype(np.int32)
    label = rng.integers(0, n_comp, n_rows)
    comp = np.empty(n_rows, dtype=np.int32)           # id of a component = its smallest frame id
    for lab in np.unique(label):
        members = np.nonzero(label == lab)[0]
        comp[members] = members.min()
    ct = torch.from_numpy(c).cuda()
    best, pops = dens.radius_min_edge(ct, r2, torch.from_numpy(comp).cuda(), torch.from_numpy(rank).cuda())
    best = best.cpu().numpy().view(np.uint64)
    want = np.full(n_rows, np.iinfo(np.uint64).max, dtype=np.uint64)
    deg = np.ones(n_rows, dtype=np.int64)
    for a, b in _brute_pairs(c, r2):
        deg[a] += 1
        deg[b] += 1
        if comp[a] == comp[b]:
            continue
        key = np.uint64((int(max(rank[a], rank[b])) << 32) | int(min(rank[a], rank[b])))
        for f in (a, b):
            want[comp[f]] = min(want[comp[f]], key)
    assert (best == want).all()
    assert (pops.cpu().numpy().astype(np.int64) == deg).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_cols,r2,n_seg", [(3000, 10, 0.045, 3), (20000, 4, 0.002, 8), (1500, 30, 0.25, 2)])
def test_min_edge_segments_merge_to_the_full_round(dens, n_rows, n_cols, r2, n_seg):
    """dc_hip_radius_min_edge_segment_dev: the candidates seen from the segments' queries merge by unsigned
    minimum (populations by summation) to those of the whole sweep; and the sharded Boruvka loop of
    clustering_amd.distributed (single rank here) returns the forest of dc_hip_radius_forest."""
    import torch
    from clustering_amd.distributed import ShardedForest
    c = gaussian_blobs(n_rows, n_cols, seed=21 + n_cols)
    rng = np.random.default_rng(n_rows + 7)
    rank = rng.permutation(n_rows).astype(np.int32)
    comp = (np.arange(n_rows) // 5 * 5).astype(np.int32)           # components of five consecutive frames
    ct, compt, rankt = torch.from_numpy(c).cuda(), torch.from_numpy(comp).cuda(), torch.from_numpy(rank).cuda()
    full_b, full_p = dens.radius_min_edge(ct, r2, compt, rankt)
    acc_b = torch.full_like(full_b, -1)
    acc_p = torch.zeros_like(full_p)
    for g in range(n_seg):
        b, p = dens.radius_min_edge(ct, r2, compt, rankt, g, n_seg)
        acc_b = torch.from_numpy(np.minimum(acc_b.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64))
                                 .view(np.int64)).cuda()
        acc_p += p
    assert bool((acc_b == full_b).all()) and bool((acc_p == full_p).all())
    edges, rounds = ShardedForest().run(ct, r2, rankt)
    want, _ = dens.radius_forest(c, r2, rank.astype(np.uint32))
    norm = lambda e: sorted((int(min(a, b)), int(max(a, b))) for a, b in e)
    assert norm(edges) == norm(want)


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_cols,r2", [(700, 3, 0.02), (3000, 10, 0.045), (1500, 30, 0.25), (257, 2, 1e-4),
                                               (20000, 4, 0.002), (1, 3, 1.0), (2, 3, 100.0)])
def test_radius_forest_has_the_connectivity_of_the_radius_graph(dens, n_rows, n_cols, r2):
    """dc_hip_radius_forest: a forest (no cycles) of pairs of the radius graph that, restricted to
    max(rank) < t, connects exactly what the whole graph restricted to max(rank) < t connects."""
    import torch
    from clustering_amd.synth import gaussian_blobs
    c = gaussian_blobs(n_rows, n_cols, seed=5 + n_cols)
    rng = np.random.default_rng(n_rows + 1)
    rank = rng.permutation(n_rows).astype(np.uint32)
    edges, rounds = dens.radius_forest(c, r2, rank)
    if n_rows <= 3000:
        all_pairs = np.array(sorted(_brute_pairs(c, r2)), dtype=np.int64).reshape(-1, 2)
    else:
        all_pairs = dens.radius_pairs(torch.from_numpy(c).cuda(), r2)[0].cpu().numpy()
    pair_set = {(int(min(a, b)), int(max(a, b))) for a, b in all_pairs}
    assert all((int(min(a, b)), int(max(a, b))) in pair_set for a, b in edges)
    full = _components(n_rows, all_pairs)
    assert len(edges) == n_rows - len(np.unique(full)), "not a spanning forest"
    w_all = np.maximum(rank[all_pairs[:, 0]], rank[all_pairs[:, 1]]) if len(all_pairs) else np.zeros(0)
    w_for = np.maximum(rank[edges[:, 0]], rank[edges[:, 1]]) if len(edges) else np.zeros(0)
    for t in [0, n_rows // 7, n_rows // 3, n_rows // 2, (3 * n_rows) // 4, n_rows]:
        a = _components(n_rows, all_pairs[w_all < t])
        b = _components(n_rows, edges[w_for < t])
        assert (a == b).all(), f"connectivity differs below rank {t}"
    assert 1 <= rounds <= 26 or n_rows <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [None, (20000, 30000)])
def test_chunked_launches_match_direct(dens, rows):
    """Launches large enough to be split along the reference axis (gridDim.y > 1: partial populations
    merged with atomicAdd, partial neighbours with the 64-bit atomicMin, incumbents published between
    chunks) must equal the direct kernels bit for bit -- on all rows and on a row range (one rank of a
    sharded run)."""
    import torch
    n = 70000
    c = gaussian_blobs(n, 4, seed=77)
    ct = torch.from_numpy(c).cuda()
    lo, hi = rows if rows else (0, n)
    radii = [0.05, 0.08]
    pp = dens.calculate_populations_partial(ct, radii, lo, hi, variant="pruned")
    pd = dens.calculate_populations_partial(ct, radii, lo, hi, variant="direct")
    assert bool((pp == pd).all())
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, radii[:1], variant="direct")[0].contiguous())
    a = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="pruned")
    b = dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")
    for x, y in zip(a, b):
        assert bool((x.view(torch.int32) == y.view(torch.int32)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_cols,n_seg", [(5000, 10, 2), (70000, 4, 8), (1000, 30, 3), (37, 2, 5), (4000, 40, 2), (3000, 70, 2)])
def test_segments_of_a_sharded_run_merge_to_the_full_result(dens, n_rows, n_cols, n_seg):
    """dc_hip_*_segment_dev: the segments of a sharded run (every n_seg-th query group of the spatial order
    with the pruned sweep, row blocks otherwise -- n_cols = 70 has no matrix-core kernel) partition the rows (of the
    neighbour sweep: every row is answered by exactly one segment): summed populations and min-merged (d2, index)
    words equal the single-device result bit for bit."""
    import torch
    c = gaussian_blobs(n_rows, n_cols, seed=5 + n_seg)
    ct = torch.from_numpy(c).cuda()
    radii = [0.1, 0.25] if n_cols < 20 else [0.5]
    full_p = dens.calculate_populations_partial(ct, radii)
    fe = dens.calculate_free_energies(full_p[0].contiguous())
    full_n = dens.nearest_neighbors_partial(ct, fe)
    acc_p = torch.zeros_like(full_p)
    words = None
    owned = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
    blocks = []
    for g in range(n_seg):
        p = dens.calculate_populations_segment(ct, radii, g, n_seg)   # partial counts (the symmetric sweep: of all rows)
        acc_p += p
        # (the second call of a populations -> neighbours pair over one array: DC_FLAG_STATS_VALID)
        a, b, cc, d = dens.nearest_neighbors_segment(ct, fe, g, n_seg, stats_valid=True)
        blocks.append(dens.pack_neighbor_block(ct, a, b, cc, d, g, n_seg))    # dc_hip_neighbors_block_pack_dev
        owned += (a <= n_rows).to(torch.int32) if n_rows > 1 else 1    # rows of other segments hold (n_rows + 1, FLT_MAX)
        w = torch.stack([(b.view(torch.int32).to(torch.int64) << 32) | (a.to(torch.int64) & 0xFFFFFFFF),
                         (d.view(torch.int32).to(torch.int64) << 32) | (cc.to(torch.int64) & 0xFFFFFFFF)])
        assert bool((dens.pack_neighbors(a, b, cc, d) == w).all())      # dc_hip_neighbors_pack_dev
        words = w if words is None else torch.minimum(words, w)
    assert bool((owned == 1).all()), "every row belongs to exactly one segment"
    assert bool((acc_p == full_p).all())
    assert bool(((words[0] & 0xFFFFFFFF).to(torch.int32) == full_n[0]).all())
    assert bool(((words[0] >> 32).to(torch.int32) == full_n[1].view(torch.int32)).all())
    assert bool(((words[1] & 0xFFFFFFFF).to(torch.int32) == full_n[2]).all())
    assert bool(((words[1] >> 32).to(torch.int32) == full_n[3].view(torch.int32)).all())
    for got, want in zip(dens.unpack_neighbors(words.contiguous()), full_n):   # dc_hip_neighbors_unpack_dev
        assert bool((got.view(torch.int32) == want.view(torch.int32)).all())
    # the all-gather merge: the segments' dense blocks, stacked as an all_gather_into_tensor would, scattered back
    rows = dens.neighbor_block_rows(n_rows, n_cols, n_seg)
    assert all(tuple(b.shape) == (4, rows) for b in blocks) and n_seg * rows >= n_rows
    H = dens.BLOCK_HEADER_ROWS    # (the layout header at the end of plane 0)
    live = sum(int((b[0, :-H] <= n_rows).sum()) for b in blocks) if n_rows > 1 else n_rows
    assert live == n_rows, "every row sits in exactly one block"
    assert all(bool((b[0, -H:-H + 8] == blocks[0][0, -H:-H + 8]).all()) for b in blocks), "one layout for all ranks"
    bad = torch.stack(blocks).contiguous()
    bad[1, 0, -H + 5] ^= 1                                  # a rank that derived another order
    keep = tuple(torch.full((n_rows,), 7, dtype=dt, device="cuda") for dt in (torch.int32, torch.float32, torch.int32, torch.float32))
    with pytest.raises(RuntimeError):
        dens.unpack_neighbor_blocks(ct, bad, n_seg, out=keep)
    if n_cols <= 64:      # (the unpack kernel itself refused the blocks: nothing was scattered into the result arrays)
        assert all(bool((t == 7).all()) for t in keep)
        dens.unpack_neighbor_blocks(ct, bad, n_seg, out=keep, check=False)     # a host that asks later ...
        assert dens.layout_status(ct.device) and all(bool((t == 7).all()) for t in keep)
    for got, want in zip(dens.unpack_neighbor_blocks(ct, torch.stack(blocks).contiguous(), n_seg), full_n):
        assert bool((got.view(torch.int32) == want.view(torch.int32)).all())   # dc_hip_neighbors_block_unpack_dev


SHARED_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
oracle = Oracle()
rng = np.random.default_rng(7)
def equals_oracle(t, want_u64, what):
    assert (t.cpu().numpy().astype(np.uint32).astype(np.uint64) == want_u64).all(), what
for n, d, r in [(9000, 30, 0.5), (5000, 24, 0.45), (3000, 40, 0.6), (20000, 10, 0.2), (700, 3, 0.1), (4097, 17, 0.35),
                (33, 30, 0.5), (1, 26, 0.5)]:
    c = gaussian_blobs(n, d, seed=n + d)
    c[rng.integers(0, n, n // 5)] = c[rng.integers(0, n, n // 5)]          # duplicates
    ct = torch.from_numpy(c).cuda()
    want = dens.calculate_populations_partial(ct, [r, 0.8 * r], variant="direct")
    got = dens.calculate_populations_partial(ct, [r, 0.8 * r], variant="pruned")
    assert bool((got == want).all()), (n, d, "all rows")
    # ... and against the ORACLE itself (not only the direct kernels): every form of this child on every shape
    want_o = oracle.populations(c, [r, 0.8 * r])
    equals_oracle(got, want_o, (n, d, "all rows vs oracle"))
    lo, hi = n // 3, n // 3 + max(1, n // 2)
    assert bool((dens.calculate_populations_partial(ct, [r], lo, hi, variant="pruned")
                 == dens.calculate_populations_partial(ct, [r], lo, hi, variant="direct")).all()), (n, d, "row range")
    acc = torch.zeros_like(want[:1])
    for g in range(3):
        acc += dens.calculate_populations_segment(ct, [r], g, 3)
    assert bool((acc == want[:1]).all()), (n, d, "segments")
    equals_oracle(acc, want_o[:1], (n, d, "segments vs oracle"))
    # several radii in ONE sweep (wide rows only: 5..8 MFMAs per chain), 2..17 radii in any order
    # (ascending radii: the symmetric sweep leaves out the leading radii a chain holds nothing of -- "wide" runs from far
    #  below to far above the typical pair distance, so that chains with and without the skip occur)
    for n_rad, order in ((2, "any"), (4, "any"), (5, "any"), (8, "any"), (9, "any"), (17, "any"), (4, "ascending"),
                         (8, "ascending"), (8, "wide"), (6, "wide")):
        radii = [float(x) for x in r * rng.uniform(0.5, 1.3, n_rad)]
        if order == "ascending":
            radii = sorted(radii)
        if order == "wide":
            radii = [float(x) for x in r * np.linspace(0.25, 1.6, n_rad)]
        want_m = dens.calculate_populations_partial(ct, radii, variant="direct")
        got_m = dens.calculate_populations_partial(ct, radii, variant="pruned")
        assert bool((got_m == want_m).all()), (n, d, n_rad, order)
        if n_rad in (4, 8, 17):
            equals_oracle(got_m, oracle.populations(c, radii), (n, d, n_rad, order, "multi-radius sweep vs oracle"))
        acc = torch.zeros_like(want_m)
        for g in range(2):
            acc += dens.calculate_populations_segment(ct, radii, g, 2)
        assert bool((acc == want_m).all()), (n, d, n_rad, "segments")
print("ok")
"""


@pytest.mark.gpu
def test_shared_operand_population_sweep():
    """pop_shared_kernel (reference operands shared through LDS by the workgroup; taken by itself only for wide
    rows and large images, e.g. C5) forced on for small shapes of every kind -- all rows, a row range, the
    segments of a sharded run, duplicates, 1..8 MFMAs per chain, and (wide rows) up to eight radii per sweep --
    against the direct kernels AND against the oracle (all rows, the segment sums and the 4 / 8 / 17-radius sweeps of
    every shape), bit for bit.  Three runs: the default (several radii in one SYMMETRIC sweep, pop_msym_kernel:
    reference-side counts through lane-private LDS accumulators), the one-sided multi-radius sweep (DC_POP_MSYM=0) and
    the round-2 symmetric form with its per-wave atomics (DC_POP_MSYM=0 DC_POP_SHARED_SYM=2)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({}, {"DC_POP_MSYM": "0"}, {"DC_POP_MSYM": "0", "DC_POP_SHARED_SYM": "2"}):
        r = subprocess.run([sys.executable, "-c", SHARED_CHILD, root], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, DC_POP_SHARED="1", **extra))
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stderr[-3000:])


INPLACE_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
oracle = Oracle()
seen = set()
for n, d, radii, in_place in [(7000, 30, [0.40, 0.45, 0.50, 0.55, 0.60], True),          # steps of a few thousand scaled units
                              (7000, 30, [0.55, 0.40, 0.60, 0.45], True),                # any order: steps of either sign
                              # (a step is at most S r2max, and the band holds that near 6.5e4 / 7.9e4 / 1.0e5 scaled units at
                              #  6 / 5 / 4 MFMAs per chain: only a radius far beyond the others AND the data's extent leaves fp16)
                              (5000, 24, [0.05, 3.0], False),
                              (4097, 17, [0.05, 3.0, 0.5], False),
                              (7000, 30, [0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 2.5], True),
                              (5000, 24, [0.30, 0.40, 0.50], True), (3000, 40, [0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2], True)]:
    c = gaussian_blobs(n, d, seed=3 * n + d)
    c[: n // 7] = c[n // 2: n // 2 + n // 7]            # duplicates: pairs at distance 0, inside every radius
    ct = torch.from_numpy(c).cuda()
    got = dens.calculate_populations_partial(ct, radii, variant="pruned")
    tiles, issued = dens.evaluated_tiles(ct.device)[0], dens.issued_mfmas(ct.device)[0]
    nm = (3 * d + 2 + 15) // 16
    assert (issued > tiles * nm) == in_place, (n, d, radii, tiles * nm, issued)   # the steps are MFMAs the kernel counts
    seen.add(issued > tiles * nm)
    want = oracle.populations(c, radii)
    assert (got.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, radii)
    acc = torch.zeros_like(got)
    for g in range(3):
        acc += dens.calculate_populations_segment(ct, radii, g, 3)
    assert (acc.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, radii, "segments")
assert seen == {True, False}
print("ok")
"""


@pytest.mark.gpu
def test_multi_radius_thresholds_in_place_and_on_the_vector_unit():
    """The symmetric multi-radius sweep takes the thresholds of radii 1 .. off the accumulator by one MFMA each (ones x
    fp16 pieces of the step; the band of the scale pays for the steps: guard_shift) -- unless a step does not fit fp16,
    then the instance that subtracts on the vector unit runs.  Both against the oracle, all rows and segment sums; which
    instance ran is read off the kernel's own count of issued MFMAs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", INPLACE_CHILD, root], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DC_POP_SHARED="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


NN_SHARED_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
oracle = Oracle()
rng = np.random.default_rng(9)
for n, d, r in [(9000, 30, 0.5), (5000, 24, 0.45), (3000, 40, 0.6), (20000, 10, 0.2), (700, 3, 0.1), (4097, 17, 0.35),
                (33, 30, 0.5), (2, 26, 0.5), (70000, 30, 0.5)]:
    c = gaussian_blobs(n, d, seed=2 * n + d)
    if n > 10:
        c[rng.integers(0, n, n // 5)] = c[rng.integers(0, n, n // 5)]      # duplicates: ties on d2, lowest index wins
    ct = torch.from_numpy(c).cuda()
    fe = dens.calculate_free_energies(dens.calculate_populations_partial(ct, [r], variant="direct")[0].contiguous())
    want = dens.nearest_neighbors_partial(ct, fe, variant="direct")
    got = dens.nearest_neighbors_partial(ct, fe, variant="pruned")
    for a, b in zip(got, want):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "all rows")
    if n <= 20000:   # ... and against the ORACLE itself: indices and d2 bits
        exp = oracle.nearest_neighbors(c, fe.cpu().numpy())
        g = [t.cpu().numpy() for t in got]
        assert (g[0].astype(np.uint32).astype(np.uint64) == exp[0]).all() and (g[2].astype(np.uint32).astype(np.uint64) == exp[2]).all(), (n, d, "idx vs oracle")
        assert (g[1].view(np.uint32) == exp[1].view(np.uint32)).all() and (g[3].view(np.uint32) == exp[3].view(np.uint32)).all(), (n, d, "d2 vs oracle")
    lo, hi = n // 3, n // 3 + max(1, n // 2)
    for a, b in zip(dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="pruned"),
                    dens.nearest_neighbors_partial(ct, fe, lo, hi, variant="direct")):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "row range")
    words = None
    for g in range(3):
        w = dens.pack_neighbors(*dens.nearest_neighbors_segment(ct, fe, g, 3))
        words = w if words is None else torch.minimum(words, w)
    for a, b in zip(dens.unpack_neighbors(words), want):
        assert bool((a.view(torch.int32) == b.view(torch.int32)).all()), (n, d, "segments")
print("ok")
"""


@pytest.mark.gpu
def test_shared_operand_neighbour_sweep():
    """nn_shared_kernel (workgroup-wide rings and survivor lists, reference operands through an LDS ring; taken by
    itself only for wide rows and large images, e.g. C5) forced on for small shapes -- all rows, a row range, the
    segments of a sharded run, duplicates (ties), 1..8 MFMAs per chain, more than one reference share -- against
    the direct kernels and (shapes up to 20 000 rows) against the oracle: indices and d2 bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", NN_SHARED_CHILD, root], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DC_NN_SHARED="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


_SWEEP_FORMS_CHILD = r"""
import sys, json
import numpy as np, torch
sys.path.insert(0, {root!r})
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
oracle = Oracle()
out = {{}}
for n, d, radii in {cases!r}:
    c = gaussian_blobs(n, d, seed=11)
    c[: n // 7] = c[n // 3: n // 3 + n // 7]          # duplicates: band pairs at distance 0 and ties
    ct = torch.from_numpy(c).cuda()
    p = dens.calculate_populations_partial(ct, radii)
    if n <= 20000:    # every form against the ORACLE (not only against each other)
        want = oracle.populations(c, radii)
        assert (p.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, "pops vs oracle")
    acc = torch.zeros_like(p)
    for g in range(3):
        acc += dens.calculate_populations_segment(ct, radii, g, 3)
    fe = dens.calculate_free_energies(p[0].contiguous())
    nn = dens.nearest_neighbors_partial(ct, fe)
    if n <= 20000:
        exp = oracle.nearest_neighbors(c, oracle.free_energies(want[0]))
        g = [t.cpu().numpy() for t in nn]
        assert (g[0].astype(np.uint32).astype(np.uint64) == exp[0]).all() and (g[2].astype(np.uint32).astype(np.uint64) == exp[2]).all(), (n, d, "nn idx vs oracle")
        assert (g[1].view(np.uint32) == exp[1].view(np.uint32)).all() and (g[3].view(np.uint32) == exp[3].view(np.uint32)).all(), (n, d, "nn d2 vs oracle")
    out[f"{{n}}x{{d}}"] = [int(p.to(torch.int64).sum()), int((p.to(torch.int64) * torch.arange(1, p.numel() + 1, device=p.device).view_as(p)).sum() % (1 << 61)),
                          bool((acc == p).all()), int(nn[0].to(torch.int64).sum()), int(nn[2].to(torch.int64).sum()),
                          int(nn[1].view(torch.int32).to(torch.int64).sum()), int(nn[3].view(torch.int32).to(torch.int64).sum())]
print("FORMS " + json.dumps(out))
"""


@pytest.mark.gpu
def test_sweep_forms_agree():
    """The population sweep has a symmetric form (every pair of query groups once, both frames credited) and a
    one-sided one, workgroups of one or four waves, and a symmetric shared-operand sweep for one or several radii; the
    neighbour sweep runs its reference shares as single waves or as the waves of one workgroup (COOP, round 6):
    every combination, in its own process (the switches are read once), reproduces the ORACLE on the shapes up to
    20 000 rows (populations of every radius, neighbour indices and d2 bits) and gives the same populations (plain and
    position-weighted checksums), the same sums over three segments, and the same neighbours -- on shapes with an even
    and an odd number of query groups, more than one reference share, 1 to 6 MFMAs per chain, and duplicated rows."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [(1152, 10, [0.2]), (1344, 10, [0.3, 0.15]), (40000, 10, [0.2]), (9000, 3, [0.05]), (20000, 16, [0.4, 0.3, 0.5]),
             (12000, 30, [0.6])]
    envs = [{}, {"DC_POP_SYM": "0"}, {"DC_WAVES_PER_GROUP": "4"}, {"DC_WAVES_PER_GROUP": "1"},
            {"DC_POP_SHARED": "1", "DC_POP_SHARED_SYM": "2"}, {"DC_POP_SHARED": "1", "DC_POP_SHARED_SYM": "0", "DC_NN_SHARED": "1"},
            # round 6: the neighbour sweep's shares as the waves of one workgroup (COOP) -- forced on, with share floors that
            # give these small shapes many reference shares -- in workgroups of 4, 2 and 8 waves
            {"DC_NN_COOP": "1", "DC_SHARE_FLOOR": "16"}, {"DC_NN_COOP": "1", "DC_SHARE_FLOOR": "40", "DC_NN_COOP_WAVES": "2"},
            {"DC_NN_COOP": "1", "DC_SHARE_FLOOR": "8", "DC_NN_COOP_WAVES": "8"}, {"DC_NN_COOP": "0", "DC_SHARE_FLOOR": "16"}]
    results = []
    for extra in envs:
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _SWEEP_FORMS_CHILD.format(root=ROOT, cases=cases)], capture_output=True,
                           text=True, timeout=600, env=env)
        assert r.returncode == 0, (extra, r.stderr[-2000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("FORMS ")][-1]
        results.append(json.loads(line[6:]))
    for extra, got in zip(envs[1:], results[1:]):
        assert got == results[0], (extra, got, results[0])
    assert all(v[2] for v in results[0].values()), "segments sum to the full populations"


@pytest.mark.gpu
def test_degenerate_inputs_pruned_equals_direct(dens):
    """Inputs at the edges of the scale rule of the matrix-core sweeps -- all rows identical (M = 0: every pair in the
    band), two far points with a tiny radius, a radius beyond everything / of 1e-30 / infinite, coordinates of 1e15 and
    1e-15, a constant and a huge column, lattices (exact ties), n = 1 and 2, 64 columns: pruned sweeps = direct kernels."""
    import torch
    rng = np.random.default_rng(5)
    n, d = 5000, 10
    two = np.concatenate([np.zeros((n // 2, d)), np.ones((n - n // 2, d)) * 1e3])
    lattice = np.stack(np.meshgrid(np.arange(20), np.arange(20), np.arange(10)), -1).reshape(-1, 3) * 0.25
    cases = [
        ("identical rows", np.full((n, d), 0.37), [0.0, 1e-3, 1.0]),
        ("all zero", np.zeros((n, d)), [0.5]),
        ("two far points, tiny radius", two, [1e-6, 10.0]),
        ("radius beyond everything", two, [1e9]),
        ("1e15", rng.normal(size=(n, d)) * 1e15, [2e15, 5e15]),
        ("1e-15", rng.normal(size=(n, d)) * 1e-15, [2e-15, 5e-15]),
        ("constant and huge column", np.concatenate([rng.normal(size=(n, d - 2)), np.full((n, 1), 7.0),
                                                     rng.normal(size=(n, 1)) * 1e6], 1), [3.0, 1e6]),
        ("radius 1e-30", rng.normal(size=(n, d)), [1e-30, 3.0]),
        ("radius inf", rng.normal(size=(300, 3)), [float("inf")]),
        ("n = 1", rng.normal(size=(1, 7)), [1.0]),
        ("n = 2 identical", np.ones((2, 40)), [0.0, 1.0]),
        ("64 columns", rng.normal(size=(3000, 64)), [8.0, 11.0, 12.5]),
        ("lattice", lattice, [0.25, 0.5, 0.3535534]),
    ]
    for name, c, radii in cases:
        ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
        a = dens.calculate_populations_partial(ct, radii, variant="pruned")
        b = dens.calculate_populations_partial(ct, radii, variant="direct")
        assert bool((a == b).all()), name
        fe = dens.calculate_free_energies(b[0].contiguous())
        for p, q in zip(dens.nearest_neighbors_partial(ct, fe, variant="pruned"),
                        dens.nearest_neighbors_partial(ct, fe, variant="direct")):
            assert bool((p.view(torch.int32) == q.view(torch.int32)).all()), name


@pytest.mark.gpu
def test_stats_valid_flag_is_checked_on_the_device(dens, oracle):
    """DC_FLAG_STATS_VALID: the neighbour call of a populations -> neighbours pair over ONE array skips the three
    statistics passes and gives the same bits; claimed for an array whose statistics the workspace does NOT hold
    (another array of the same shape went through last, a direct-variant call that left no statistics, or the SAME
    buffer rewritten in place: the guard compares address, shape and a content fingerprint) the check on the device
    flags the sweep and the direct kernels answer: slower, never different."""
    import torch
    n, d = 6000, 10
    c1 = gaussian_blobs(n, d, seed=41)
    c2 = (gaussian_blobs(n, d, seed=42) * 37.0 + 5.0).astype(np.float32)     # other scale: stale statistics would hurt
    t1, t2 = torch.from_numpy(c1).cuda(), torch.from_numpy(c2).cuda()

    def reference(c):
        pops = oracle.populations(c, [7.0 if c is c2 else 0.2])
        fe = oracle.free_energies(pops[0])
        return pops, fe, oracle.nearest_neighbors(c, fe)

    def same(nn, exp):
        g = [t.cpu().numpy() for t in nn]
        return ((g[0].astype(np.uint32).astype(np.uint64) == exp[0]).all() and (g[2].astype(np.uint32).astype(np.uint64) == exp[2]).all()
                and (bits(g[1]) == bits(exp[1])).all() and (bits(g[3]) == bits(exp[3])).all())

    p1, fe1, nn1 = reference(c1)
    p2, fe2, nn2 = reference(c2)
    f1, f2 = torch.from_numpy(fe1).cuda(), torch.from_numpy(fe2).cuda()
    # the legitimate pair
    got = dens.calculate_populations_partial(t1, [0.2])
    assert (got.cpu().numpy().astype(np.uint32).astype(np.uint64) == p1).all()
    assert same(dens.nearest_neighbors_partial(t1, f1, stats_valid=True), nn1)
    assert dens.evaluated_tiles(t1.device)[1] > 0                      # the pruned matrix-core sweep ran
    # a false claim: the workspace holds the statistics of c1, the call is about c2
    assert same(dens.nearest_neighbors_partial(t2, f2, stats_valid=True), nn2)
    assert dens.evaluated_tiles(t2.device)[1] == 0                     # flagged on the device: the direct kernel answered
    assert (dens.calculate_populations_partial(t2, [7.0], stats_valid=True).cpu().numpy().astype(np.uint32).astype(np.uint64) == p2).all()
    # ... and after a direct-variant call (no statistics at all in the header)
    dens.calculate_populations_partial(t2, [7.0])                      # (fresh statistics of c2)
    assert same(dens.nearest_neighbors_partial(t2, f2, stats_valid=True), nn2)
    assert dens.evaluated_tiles(t2.device)[1] > 0
    # the same buffer REWRITTEN IN PLACE (same address, same shape, other contents): the fingerprint catches it
    t2.copy_(t1)
    assert same(dens.nearest_neighbors_partial(t2, f1, stats_valid=True), nn1)
    assert dens.evaluated_tiles(t2.device)[1] == 0                     # flagged: the statistics were c2's
    assert (dens.calculate_populations_partial(t2, [0.2], stats_valid=True).cpu().numpy().astype(np.uint32).astype(np.uint64) == p1).all()
    assert dens.evaluated_tiles(t2.device)[0] == 0
    dens.calculate_populations_partial(t2, [0.2])                      # (fresh statistics of the new contents)
    assert same(dens.nearest_neighbors_partial(t2, f1, stats_valid=True), nn1)
    assert dens.evaluated_tiles(t2.device)[1] > 0
    # one element changed by one ulp
    t2.view(torch.int32)[4321, 7] += 1
    c3 = t2.cpu().numpy()
    p3, fe3, nn3 = reference(c3)
    assert same(dens.nearest_neighbors_partial(t2, torch.from_numpy(fe3).cuda(), stats_valid=True), nn3)
    assert dens.evaluated_tiles(t2.device)[1] == 0
