"""GPU: the libraries built for the summation order of an AVX build of the reference (`make CANON=avx`,
clustering_amd/lib_avx/libdcdensity.so, DC_CANON_ORDER=avx) and of a -march=native build on an AVX2 + FMA host (`make
CANON=fma`, lib_fma/, DC_CANON_ORDER=fma) against the oracle of the same order (oracle/dc_oracle.c with -DDCO_CANON_AVX /
-DDCO_CANON_FMA, pinned in tests/test_oracle.py against g++ -mavx / -mavx2 -mfma on the reference's loop shape,
CMakeLists.txt:53-56, 73-76):
populations, free energies, nn / nn_hd with their d2 bits through every variant -- the exact kernels, the matrix-core
sweeps with their canonical re-checks (pruned, unpruned, fp32-input), segments of a sharded run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from clustering_amd import capi, density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
ORDER = sys.argv[2]
assert capi.lib.dc_hip_canon_order().decode() == ORDER
o, o_def = Oracle(order=ORDER), Oracle()
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
differ = 0
for n, d, radii in [(3000, 3, [0.05]), (2500, 5, [0.1, 0.2]), (4000, 8, [0.2]), (4000, 9, [0.2, 0.15]), (6000, 10, [0.2, 0.25, 0.3]),
                    (3000, 12, [0.3]), (2500, 16, [0.4, 0.3]), (2000, 30, [0.6]), (1200, 70, [1.0])]:
    c = gaussian_blobs(n, d, seed=900 + d)
    c[: n // 9] = c[n // 3: n // 3 + n // 9]          # duplicates: ties, band pairs at distance 0
    ct = torch.from_numpy(c).cuda()
    want = o.populations(c, radii)
    differ += int((want != o_def.populations(c, radii)).sum())
    fe_want = o.free_energies(want[0])
    exp = o.nearest_neighbors(c, fe_want)
    variants = ["direct", "auto", "mfma"] + (["mfma32"] if d in (9, 10) else [])
    if d > 64: variants = ["direct", "auto"]
    for v in variants:
        p = dens.calculate_populations_partial(ct, radii, variant=v)
        assert (p.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, v, "pops")
        fe = dens.calculate_free_energies(p[0].contiguous())
        assert (bits(fe.cpu().numpy()) == bits(fe_want)).all(), (n, d, v, "fe")
        g = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fe, variant=v)]
        assert (g[0].astype(np.uint32).astype(np.uint64) == exp[0]).all() and (g[2].astype(np.uint32).astype(np.uint64) == exp[2]).all(), (n, d, v, "nn idx")
        assert (bits(g[1]) == bits(exp[1])).all() and (bits(g[3]) == bits(exp[3])).all(), (n, d, v, "nn d2")
    acc = torch.zeros_like(p)
    for s in range(3):
        acc += dens.calculate_populations_segment(ct, radii, s, 3)
    assert (acc.cpu().numpy().astype(np.uint32).astype(np.uint64) == want).all(), (n, d, "segments")
print("ok: populations under the two orders differ in", differ, "entries")
"""


@pytest.mark.parametrize("order", ["avx", "fma"])
def test_other_order_library_against_the_oracle_of_that_order(order):
    if not os.path.exists(os.path.join(ROOT, "clustering_amd", "lib_" + order, "libdcdensity.so")):
        pytest.fail(f"clustering_amd/lib_{order}/libdcdensity.so is missing: __graft_entry__.build() makes it")
    env = dict(os.environ, DC_CANON_ORDER=order)
    env.pop("DC_LIB_PATH", None)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, order], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]
