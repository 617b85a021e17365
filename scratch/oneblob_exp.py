"""Per-tile-pair cost of the sweeps on ONE centred blob (M = max |x - mean|^2 is the blob's own extent) against the three
blobs of C3 (same sigma, same r; M is set by the distance of the blobs from the common mean): what local origins
would buy."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
r, d = 0.2, 10
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def run(name, c):
    n = c.shape[0]
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda()
    ts = []
    for rep in range(3):
        ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    pt = dens.evaluated_tiles(ct.device)[0]
    fe = dens.calculate_free_energies(p[0].contiguous())
    tn = []
    for rep in range(3):
        ev[0].record(); dens.nearest_neighbors_partial(ct, fe); ev[1].record(); torch.cuda.synchronize()
        tn.append(ev[0].elapsed_time(ev[1]))
    nt = dens.evaluated_tiles(ct.device)[1]
    print(f"{name:28s} pop {min(ts):7.2f} ms = {min(ts)*1e6/pt*1024:6.1f} ns per tile pair per SIMD; nn {min(tn):7.2f} ms = {min(tn)*1e6/nt*1024:6.1f}; mean pop {float(p[0].float().mean()):.0f}")
c3 = gaussian_blobs(1_000_000, d)
run("three blobs (C3)", c3)
rng = np.random.default_rng(5)
one = rng.normal(0.0, 0.08, (333_333, d)).astype(np.float32)
run("one centred blob, 333k", one)
run("the same blob at offset 1", one + np.float32(1.0) * np.eye(1, d, 0, dtype=np.float32) + 0*one)
