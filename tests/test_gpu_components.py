"""Components of the pruned population sweeps (-m gpu): frames that lie far apart in columns 0/1 are measured from
their own origins (the guard band of the Gram form follows the extent of a cluster, not of the data set), clusters
that come closer than the largest radius exchange their few cross pairs through an exact kernel.  Everything against
the ORACLE: spread-out data at size (row ranges), data whose extent is 10^3 .. 10^5 radii, adjacent clusters (all
rows, row ranges, segments, several radii, the shared-operand sweeps), more clusters than component slots."""
import os
import subprocess
import sys

import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dens():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from clustering_amd import density
    return density


def spread_blobs(n, d, factor, seed=20240):
    """the bench generator's blobs with their centres moved apart by `factor` (sigma stays 0.08)"""
    base = gaussian_blobs(n, d, seed=seed)
    labels = np.random.default_rng(seed).integers(0, 3, n)      # (the generator's own first draw)
    centres = np.zeros((3, d), dtype=np.float32)
    centres[:, :2] = [(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)]
    return np.ascontiguousarray(base + (factor - 1.0) * centres[labels], dtype=np.float32)


def check_rows(dens, oracle, c, radii, ranges, want_components=None):
    """populations of every radius + neighbours of the row ranges against the oracle (all references)"""
    import torch
    ct = torch.from_numpy(c).cuda()
    pops = dens.calculate_populations_partial(ct, radii, variant="pruned")
    info = dens.components_info(ct)
    if want_components is not None:
        assert info["n_components"] == want_components, info
    got = pops.cpu().numpy().astype(np.uint32)
    for lo, hi in ranges:
        want = oracle.populations(c, radii, lo, hi)
        assert (got[:, lo:hi].astype(np.uint64) == want[:, lo:hi]).all(), (lo, hi)
    # free energies need every population: the direct kernels (oracle-checked elsewhere) and the sweep must agree on all rows
    assert bool((pops == dens.calculate_populations_partial(ct, radii, variant="direct")).all())
    fe = dens.calculate_free_energies(pops[0].contiguous())
    nn = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fe, variant="pruned")]
    fe_h = fe.cpu().numpy()
    for lo, hi in ranges:
        exp = oracle.nearest_neighbors(c, fe_h, lo, hi)
        assert (nn[0][lo:hi].astype(np.uint32).astype(np.uint64) == exp[0][lo:hi]).all(), (lo, hi)
        assert (nn[2][lo:hi].astype(np.uint32).astype(np.uint64) == exp[2][lo:hi]).all(), (lo, hi)
        assert (bits(nn[1][lo:hi]) == bits(exp[1][lo:hi])).all() and (bits(nn[3][lo:hi]) == bits(exp[3][lo:hi])).all()
    return info


@pytest.mark.parametrize("factor", [10.0, 100.0])
def test_spread_blobs_against_the_oracle_at_size(dens, oracle, factor):
    """1M x 10 with the blob centres x10 / x100 apart (extent / radius 120 / 1200): the pruned sweeps against the oracle
    on three row ranges against all 10^6 references; every cluster is a component with its own origin, and the band of
    the sweep follows the clusters' extent (extent2_local), not the data set's (extent2_global)"""
    c = spread_blobs(1_000_000, 10, factor)
    info = check_rows(dens, oracle, c, [0.2], [(0, 96), (499_000, 499_096), (999_904, 1_000_000)], want_components=3)
    assert info["extent2_local"] < 1.0 < 50.0 < info["extent2_global"]


def test_extent_of_a_thousand_and_more_radii(dens, oracle):
    """data whose extent is 10^3 .. 10^5 times the radius: far clusters (components), one far outlier frame, and a
    uniform cloud (one component: the band is wide, the results still exact)"""
    rng = np.random.default_rng(3)
    n, d = 60000, 6
    clusters = np.concatenate([rng.normal(size=(n // 3, d)) * 0.05 + off for off in (0.0, 300.0, -7000.0)]).astype(np.float32)
    clusters = clusters[rng.permutation(len(clusters))]
    check_rows(dens, oracle, clusters, [0.1, 0.07], [(0, 400), (30000, 30400)], want_components=3)
    outlier = clusters.copy()
    outlier[:, 0] = np.where(np.arange(len(outlier)) == 17, 9.0e4, outlier[:, 0])
    check_rows(dens, oracle, outlier, [0.1], [(0, 400)])
    cloud = (rng.uniform(size=(n, d)) * np.array([1000.0, 1000.0, 1, 1, 1, 1])).astype(np.float32)
    info = check_rows(dens, oracle, cloud, [1.0], [(0, 400), (59000, 59400)])
    assert info["n_components"] >= 1


def adjacent_clusters(n, d, gap_over_r, r, seed):
    """two clusters whose 2-D projections are gap_over_r * r apart at their closest frames (between r / 2 and r: two
    components that are ADJACENT -- cross pairs exist), plus a third far away"""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n // 2, d)).astype(np.float32) * 0.05
    b = rng.normal(size=(n - n // 2 - n // 10, d)).astype(np.float32) * 0.05
    far = rng.normal(size=(n // 10, d)).astype(np.float32) * 0.05 + 6.0
    shift = (a[:, 0].max() - b[:, 0].min()) + gap_over_r * r
    b[:, 0] += shift
    c = np.concatenate([a, b, far])
    return np.ascontiguousarray(c[rng.permutation(len(c))], dtype=np.float32)


@pytest.mark.parametrize("n_cols,radii", [(5, [0.3]), (10, [0.3, 0.2, 0.25]), (3, [0.3])])
def test_adjacent_components_exchange_their_cross_pairs(dens, oracle, n_cols, radii):
    """gap = 0.7 r_max: the clusters are separate components (connectivity r_max / 2) but closer than r_max -- the pairs
    between them come from pop_cross_kernel.  All rows, a row range, three segments, against the oracle."""
    import torch
    c = adjacent_clusters(24000, n_cols, 0.7, max(radii), seed=n_cols)
    ct = torch.from_numpy(c).cuda()
    want = oracle.populations(c, radii)
    got = dens.calculate_populations_partial(ct, radii, variant="pruned")
    assert dens.components_info(ct)["n_components"] == 3
    assert (got.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all()
    # there really are cross pairs: with the clusters pulled apart the populations differ
    lo, hi = 5000, 17000
    part = dens.calculate_populations_partial(ct, radii, lo, hi, variant="pruned").cpu().numpy().astype(np.uint32)
    assert (part[:, lo:hi].astype(np.uint64) == want[:, lo:hi]).all() and not part[:, :lo].any()
    acc = torch.zeros_like(got)
    for g in range(3):
        acc += dens.calculate_populations_segment(ct, radii, g, 3)
    assert (acc.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all()
    far = c.copy()
    far[c[:, 0] > np.median(c[:, 0]), 0] += 10.0
    assert (oracle.populations(far, radii[:1]) != want[:1]).any(), "the test data has no cross pairs"
    # neighbours: the frames at the facing edges may have their (lower-free-energy) neighbour in the other cluster, the
    # lowest free energy of a cluster always has -- the exact cross pass (nn_cross_kernel); all rows, a row range, segments
    fe = oracle.free_energies(want[0])
    exp = oracle.nearest_neighbors(c, fe)
    fe_t = torch.from_numpy(fe).cuda()

    def same(nn):
        g = [t.cpu().numpy() for t in nn]
        return ((g[0].astype(np.uint32).astype(np.uint64) == exp[0]).all() and (g[2].astype(np.uint32).astype(np.uint64) == exp[2]).all()
                and (bits(g[1]) == bits(exp[1])).all() and (bits(g[3]) == bits(exp[3])).all())
    assert same(dens.nearest_neighbors_partial(ct, fe_t, variant="pruned"))
    assert same(dens.nearest_neighbors_partial(ct, fe_t, variant="pruned", stats_valid=True))   # (the partition of the last sweep)
    part = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fe_t, lo, hi, variant="pruned")]
    assert (part[0][lo:hi].astype(np.uint32).astype(np.uint64) == exp[0][lo:hi]).all()
    assert (part[2][lo:hi].astype(np.uint32).astype(np.uint64) == exp[2][lo:hi]).all()
    assert (bits(part[1][lo:hi]) == bits(exp[1][lo:hi])).all() and (bits(part[3][lo:hi]) == bits(exp[3][lo:hi])).all()
    words = None
    for g in range(3):
        w = dens.pack_neighbors(*dens.nearest_neighbors_segment(ct, fe_t, g, 3))
        words = w if words is None else torch.minimum(words, w)
    assert same(dens.unpack_neighbors(words.contiguous()))
    comp_of = (c[:, 0] > 3.0).astype(int) * 2 + ((c[:, 0] > (c[c[:, 0] < 3.0, 0].min() + c[c[:, 0] < 3.0, 0].max()) / 2) & (c[:, 0] < 3.0))
    crossing = comp_of[exp[2][exp[2] < len(c)].astype(int)] != comp_of[np.nonzero(exp[2] < len(c))[0]]
    assert crossing.any(), "no lower-free-energy neighbour crosses between the clusters in this data"


_SHARED_ADJ_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from clustering_amd import density as dens
from oracle.oracle import Oracle
from test_gpu_components import adjacent_clusters
o = Oracle()
for n, d, radii in [(12000, 30, [0.55, 0.6, 0.45, 0.65, 0.5, 0.4, 0.62, 0.58]), (12000, 30, [0.65]), (9000, 16, [0.5, 0.4, 0.3])]:
    c = adjacent_clusters(n, d, 0.8, max(radii), seed=d)
    ct = torch.from_numpy(c).cuda()
    want = o.populations(c, radii)
    got = dens.calculate_populations_partial(ct, radii, variant="pruned")
    assert dens.components_info(ct)["n_components"] == 3, dens.components_info(ct)
    assert (got.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, "all rows")
    acc = torch.zeros_like(got)
    for g in range(2):
        acc += dens.calculate_populations_segment(ct, radii, g, 2)
    assert (acc.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, "segments")
    lo, hi = n // 4, n // 4 + n // 3
    part = dens.calculate_populations_partial(ct, radii, lo, hi, variant="pruned").cpu().numpy().astype(np.uint32)
    assert (part[:, lo:hi].astype(np.uint64) == want[:, lo:hi]).all(), (n, d, "row range")
print("ok")
"""


def test_adjacent_components_in_the_shared_operand_sweeps():
    """the same through pop_shared_kernel (one radius: symmetric; eight radii in one sweep; forced on for a small shape)"""
    for extra in ({"DC_POP_SHARED": "1"}, {"DC_POP_SHARED": "1", "DC_POP_SHARED_SYM": "2"}):
        r = subprocess.run([sys.executable, "-c", _SHARED_ADJ_CHILD, ROOT], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **extra))
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stderr[-3000:])


def test_more_clusters_than_component_slots(dens, oracle):
    """100 small clusters on a lattice (more than the 64 components the sweep keeps apart): one component, one origin --
    the old sweep -- and exact; 36 clusters: 36 components"""
    import torch
    rng = np.random.default_rng(11)
    for k, want_c in ((10, 1), (6, 36)):
        centres = np.stack(np.meshgrid(np.arange(k), np.arange(k)), -1).reshape(-1, 2) * 5.0
        n_per = 300
        c = np.zeros((k * k * n_per, 4), dtype=np.float32)
        c[:, :2] = np.repeat(centres, n_per, axis=0)
        c += rng.normal(size=c.shape).astype(np.float32) * 0.03
        c = np.ascontiguousarray(c[rng.permutation(len(c))])
        ct = torch.from_numpy(c).cuda()
        got = dens.calculate_populations_partial(ct, [0.08, 0.04], variant="pruned")
        assert dens.components_info(ct)["n_components"] == want_c
        assert (got.cpu().numpy().astype(np.uint32).astype(np.uint64) == oracle.populations(c, [0.08, 0.04])).all()


_OFF_CHILD = r"""
import sys, json
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from clustering_amd import density as dens
from test_gpu_components import spread_blobs
c = spread_blobs(300000, 10, 10.0)
ct = torch.from_numpy(c).cuda()
p = dens.calculate_populations_partial(ct, [0.2, 0.1])
print("OUT " + json.dumps({"n": dens.components_info(ct)["n_components"], "sum": int(p.to(torch.int64).sum()),
                           "w": int((p.to(torch.int64) * torch.arange(1, p.numel() + 1, device=p.device).view_as(p)).sum() % (1 << 61))}))
"""


def test_components_switched_off_give_the_same_populations():
    """DC_POP_COMPONENTS=0 (one component, one origin: the round-2 sweep) against the default on spread-out data"""
    import json
    out = []
    for extra in ({}, {"DC_POP_COMPONENTS": "0"}):
        r = subprocess.run([sys.executable, "-c", _OFF_CHILD, ROOT], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("OUT ")][-1][4:]))
    assert out[0]["n"] == 3 and out[1]["n"] == 1
    assert out[0]["sum"] == out[1]["sum"] and out[0]["w"] == out[1]["w"]
