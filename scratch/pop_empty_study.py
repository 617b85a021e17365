"""Would an early-out pay in the POPULATION sweep?  For sampled query groups (6 tiles of the population order: frames by
2-D cell of ~64 frames of the bounding box, no order inside a cell) the share of the evaluated tile pairs (tile boxes closer
than r) that hold NO pair with d2 < r^2 + skip bound (M 2^-9, M = 0.29: the extent of a C3 component) -- those could stop
after the coarse MFMA with a tile minimum instead of the counting epilogue."""
import sys
import numpy as np
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
n, d, TQ, r = 1_000_000, 10, 6, 0.2
c = gaussian_blobs(n, d)
rng = np.random.default_rng(3)
lo, hi = c[:, :2].min(0), c[:, :2].max(0)
edge = np.sqrt(np.prod(hi - lo) * 64.0 / n)
ix = ((c[:, 0] - lo[0]) / edge).astype(np.int64); iy = ((c[:, 1] - lo[1]) / edge).astype(np.int64)
nby = iy.max() + 1
iy_s = np.where(ix & 1, nby - 1 - iy, iy)
order = np.argsort(ix * nby + iy_s, kind='stable')
cs = c[order]
T = n // 32
blo = cs[:T * 32].reshape(T, 32, d)[:, :, :2].min(1); bhi = cs[:T * 32].reshape(T, 32, d)[:, :, :2].max(1)
sq = (cs * cs).sum(1)
thr = r * r + 0.29 / 512 + 1e-4
empty = []; eva = []
for g in rng.choice(T // TQ, 40, replace=False):
    t0 = g * TQ
    q = cs[t0 * 32:(t0 + TQ) * 32]
    qlo, qhi = blo[t0:t0 + TQ].min(0), bhi[t0:t0 + TQ].max(0)
    gap = np.maximum(0.0, np.maximum(qlo - bhi, blo - qhi))
    near = np.flatnonzero((gap * gap).sum(1) < r * r)
    ref = cs[:T * 32].reshape(T, 32, d)[near]                       # [K, 32, d]
    d2 = (q * q).sum(1)[None, :, None] + (ref * ref).sum(2)[:, None, :] - 2.0 * np.einsum('qd,krd->kqr', q, ref)
    d2 = d2.reshape(len(near), TQ, 32, 32)
    m = d2.min(axis=(2, 3))                                         # [K, TQ] chain minima
    empty.append(((m > thr).mean(), (m > thr).all(1).mean())); eva.append(len(near) / T)
e = np.array(empty)
print(f"evaluated reference tiles per group: {np.mean(eva):.3f} of all; chains with no pair inside r (+ skip bound): {e[:,0].mean():.3f} "
      f"(min {e[:,0].min():.3f}, max {e[:,0].max():.3f}); reference tiles where all {TQ} chains are empty: {e[:,1].mean():.3f}")
