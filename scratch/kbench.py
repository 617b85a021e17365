import sys, time, argparse
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=200000)
ap.add_argument('--d', type=int, default=10)
ap.add_argument('--variant', default='mfma')
ap.add_argument('--radii', type=float, nargs='+', default=[0.2])
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--what', default='pop,nn')
ap.add_argument('--rows', type=int, nargs=2, default=None, help='query row range (emulates one rank of a sharded run)')
a = ap.parse_args()
c = torch.from_numpy(gaussian_blobs(a.n, a.d)).cuda()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lo, hi = (a.rows if a.rows else (0, a.n))
pops = dens.calculate_populations_partial(c, a.radii, variant=a.variant)
fe = dens.calculate_free_energies(pops[0].contiguous())
torch.cuda.synchronize()
for what in a.what.split(','):
    ts = []
    for _ in range(a.reps):
        ev0.record()
        if what == 'pop':
            dens.calculate_populations_partial(c, a.radii, lo, hi, variant=a.variant)
        else:
            dens.nearest_neighbors_partial(c, fe, lo, hi, variant=a.variant)
        ev1.record(); torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1))
    t = min(ts) * 1e-3
    tiles = dens.evaluated_tiles(c.device)[0 if what == 'pop' else 1]
    frac = tiles * 1024.0 / (float(hi - lo) * a.n) if tiles else 1.0
    print(f"   raw counter {tiles}")
    print(f"   evaluated fraction {frac:.3f} -> {frac*(hi-lo)*a.n*2*a.d/t/157.3e12*100:.1f}% fp32 roof on evaluated pairs")
    print(f"{what} {a.variant} n={a.n} d={a.d} radii={len(a.radii)}: {min(ts):.2f} ms  {a.n*a.n/t:.3e} pairs/s  {a.n*a.n*2*a.d/t/157.3e12*100:.1f}% fp32 roof")
