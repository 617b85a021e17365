"""Population sweep at C3: of the (query tile, reference tile) chains a group of six query tiles runs against the reference
tiles its GROUP box admits (gap < r), how many does the query tile's own box admit?  (2-D cells of 64 frames per component.)"""
import numpy as np, sys
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
c = gaussian_blobs(n, d)
cent = np.array([(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)], dtype=np.float32); sig = 0.08
lab = np.argmin(((c[:, None, :2] - cent[None]) ** 2).sum(2), 1)
keys = np.zeros(n, dtype=np.int64)
for k in range(3):
    m = lab == k; x = c[m]; lo = x[:, :2].min(0)
    edge = np.sqrt(np.pi * (2.5 * sig) ** 2 / (m.sum() / 64))
    keys[m] = (k * 4096 + ((x[:, 0] - lo[0]) / edge).astype(np.int64)) * 4096 + ((x[:, 1] - lo[1]) / edge).astype(np.int64)
order = np.argsort(keys, kind='stable')
cs = c[order]; T = n // 32; TQ = 6
lo = cs[:T * 32].reshape(T, 32, d).min(1)[:, :2]; hi = cs[:T * 32].reshape(T, 32, d).max(1)[:, :2]
rng = np.random.default_rng(5)
tot = kept = 0
for g in rng.choice(T // TQ, 400, replace=False):
    t0 = g * TQ
    qlo = lo[t0:t0 + TQ].min(0); qhi = hi[t0:t0 + TQ].max(0)
    gg = np.maximum(0, np.maximum(qlo - hi, lo - qhi)); listed = np.flatnonzero((gg * gg).sum(1) < r * r)
    for k in range(TQ):
        tg = np.maximum(0, np.maximum(lo[t0 + k] - hi[listed], lo[listed] - hi[t0 + k]))
        kept += ((tg * tg).sum(1) < r * r).sum()
    tot += TQ * len(listed)
print(f"chains admitted by the group box: {tot}; by the tile's own box: {kept} ({kept/tot:.4f})")
