"""Single-sort 3-D cell keys (2-D cells of columns 0/1, uniform bins of column 2) against the 2-D cells."""
import numpy as np, sys
sys.path.insert(0, '.')
sys.argv = [sys.argv[0], '1000000', '10', '0.2']
exec(open('scratch/prune_study2.py').read().split("mn = c.min(0)")[0])
mn, mx = c.min(0), c.max(0)
for cell, nz in ((0.02, 1), (0.06, 8), (0.08, 16), (0.11, 32), (0.16, 64), (0.08, 32), (0.11, 16)):
    bx = np.floor((c[:, 0] - mn[0]) / cell).astype(np.int64); by = np.floor((c[:, 1] - mn[1]) / cell).astype(np.int64)
    bz = np.minimum((nz * (c[:, 2] - mn[2]) / (mx[2] - mn[2])).astype(np.int64), nz - 1)
    key = (bx * 100000 + by) * nz + bz
    o = np.argsort(key, kind='stable')
    study(f"cells {cell} x {nz} uniform bins of col 2, 3-D boxes", o, [0, 1, 2], nq=40)
