"""the host-pointer entry point dc_hip_density_all at C3 (1M x 10, r = 0.2): wall time of the whole call with host buffers
(upload of the coordinates, the sweeps, download of populations, free energies and the four neighbour arrays) -- the
PCIe-inclusive rate next to bench.py's resident one"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
from clustering_amd import capi
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = gaussian_blobs(n, d)
radii = np.array([0.2], dtype=np.float32)
pops = np.zeros(n, dtype=np.uint32); fe = np.zeros(n, dtype=np.float32)
nn_idx = np.zeros(n, dtype=np.uint32); nn_d2 = np.zeros(n, dtype=np.float32); hd_idx = np.zeros(n, dtype=np.uint32); hd_d2 = np.zeros(n, dtype=np.float32)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    capi.check(capi.lib.dc_hip_density_all(vp(c), n, d, vp(radii), 1, 0, 1, vp(pops), vp(fe), vp(nn_idx), vp(nn_d2), vp(hd_idx), vp(hd_d2)), "dc_hip_density_all")
    ts.append((time.perf_counter() - t0) * 1e3)
print('ms per call', [round(t, 2) for t in ts], 'best', min(ts), 'frame-pairs/s (2 N^2 / t)', 2.0 * n * n / (min(ts) * 1e-3), 'mean pop', float(pops.mean()), 'sigma2-ish', float(nn_d2.astype(np.float64).mean()))
