"""A/B of the population sweep on C3: run with PYTHONPATH pointing at the package to test"""
import sys, numpy as np, torch
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
ct = torch.from_numpy(gaussian_blobs(n, d)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
dens.sweep_timing(True)
ts, ks = [], []
for rep in range(5):
    ev[0].record(); p = dens.calculate_populations_partial(ct, [r]); ev[1].record(); torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1])); ks.append(dens.last_sweep_ms("pop", ct.device))
print(dens.__file__, f"call {min(ts):.2f} ms kernel {min(ks):.2f} ms tiles {dens.evaluated_tiles(ct.device)[0]} sum {int(p.sum())}")
