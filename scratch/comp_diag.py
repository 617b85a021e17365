"""components of the pruned population sweep on the C3 data (and spread variants): count, extents, time, tiles"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
base = gaussian_blobs(n, d)
rng = np.random.default_rng(20240)
labels = rng.integers(0, 3, n)
centres = np.zeros((3, d), dtype=np.float32)
centres[:, :2] = [(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
dens.sweep_timing(True)
for f in [float(a) for a in sys.argv[1:]] or [1.0]:
    c = base + (f - 1.0) * centres[labels]
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    ts, ks = [], []
    for rep in range(4):
        ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1])); ks.append(dens.last_sweep_ms("pop", ct.device))
    print(f"x{f}: call {min(ts):.2f} ms kernel {min(ks):.2f} ms tiles {dens.evaluated_tiles(ct.device)[0]}", dens.components_info(ct))
