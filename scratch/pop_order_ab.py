"""(experiment: a -DDC_WAVE_STAMPS build with scratch/r5_pop_order_hack.patch applied) the population sweep with its query groups launched heavy-first inside every
XCD's eighth of every reference share, weights = the groups' MEASURED wave times of a first run: what the order is worth
with the caches in the loop (scratch/sched_study.py says -3.5 % for all rows of C3, -2 % for an eighth, off line)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens, capi
from clustering_amd.synth import gaussian_blobs
n, d, G = 1000000, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 1
seg = 3 if G > 1 else 0
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
dens.sweep_timing(True)
def run():
    dens.calculate_populations_segment(c, [0.2], seg, G) if G > 1 else dens.calculate_populations_partial(c, [0.2])
    torch.cuda.synchronize()
    return dens.last_sweep_ms("pop", c.device)
def set_perm(p):
    p = np.ascontiguousarray(p, dtype=np.uint32)
    assert capi.lib.dc_dbg_set_pop_perm(p.ctypes.data_as(C.c_void_p), C.c_uint32(len(p))) == 0
set_perm(np.zeros(0))
run(); kms = run()
N = 1 << 17
buf = np.zeros((N, 3), dtype=np.uint64)
assert capi.lib.dc_dbg_pop_wave_times(buf.ctypes.data_as(C.c_void_p), C.c_size_t(N)) == 0
live = buf[:, 1] > 0
live &= buf[:, 0].astype(np.int64) >= buf[:, 1].astype(np.int64).max() - int(kms * 1.3e5)
dur = (buf[live, 1].astype(np.int64) - buf[live, 0].astype(np.int64)) / 100.0
grp = (buf[live, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
groups = np.unique(grp)
# unit of a group in its segment: group // G for the cyclic deal of single groups (seg_unit with block 1)
unit = groups // G
n_units = int(unit.max()) + 1
w = np.zeros(n_units)
for g_, u_ in zip(groups, unit): w[u_] = dur[grp == g_].sum()
n_full, rem = n_units >> 3, n_units & 7
perm = np.arange(n_units, dtype=np.uint32)
for e in range(8):
    base = e * n_full + min(e, rem); cnt = n_full + (1 if e < rem else 0)
    perm[base:base + cnt] = base + np.argsort(-w[base:base + cnt], kind='stable')
ident, heavy = [], []
for rep in range(6):
    set_perm(np.zeros(0)); ident.append(run())
    set_perm(perm); heavy.append(run())
p1 = dens.calculate_populations_segment(c, [0.2], seg, G)[0] if G > 1 else dens.calculate_populations_partial(c, [0.2])[0]
set_perm(np.zeros(0))
p0 = dens.calculate_populations_segment(c, [0.2], seg, G)[0] if G > 1 else dens.calculate_populations_partial(c, [0.2])[0]
print(f"G = {G}: {n_units} units; kernel ms as launched {np.round(ident, 3).tolist()} -> heavy first inside the eighths {np.round(heavy, 3).tolist()}; "
      f"means {np.mean(ident):.3f} -> {np.mean(heavy):.3f}; same counts: {bool((p0 == p1).all())}")
