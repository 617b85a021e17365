"""Neighbour sweep: how much of a query group's ring is owed to its few worst queries?  Tile pairs to evaluate when the
confirming radius is the group's worst query (now) / the k-th worst (the k-1 worst finished by an exact follow-up), and
the tiles those left-over queries would have to search exactly.  C3 data, 2-D cells of 128 frames with free energy inside."""
import numpy as np, sys
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = gaussian_blobs(n, d)
cent = np.array([(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)], dtype=np.float32); sig = 0.08
dens = np.zeros(n)
for k in range(3):
    mu = np.zeros(d, dtype=np.float32); mu[:2] = cent[k]
    dens += np.exp(-((c - mu) ** 2).sum(1) / (2 * sig * sig))
fe = -np.log(dens + 1e-300)
lab = np.argmin(((c[:, None, :2] - cent[None]) ** 2).sum(2), 1)
keys = np.zeros(n, dtype=np.int64)
for k in range(3):
    m = lab == k; x = c[m]; lo = x[:, :2].min(0)
    edge = np.sqrt(np.pi * (2.5 * sig) ** 2 / (m.sum() / 128))
    bx = ((x[:, 0] - lo[0]) / edge).astype(np.int64); by = ((x[:, 1] - lo[1]) / edge).astype(np.int64)
    f = fe[m]; fq = ((f - f.min()) / (f.max() - f.min()) * 255).astype(np.int64)
    keys[m] = ((k * 4096 + bx) * 4096 + by) * 256 + fq
order = np.argsort(keys, kind='stable')
cs = c[order]; fs = fe[order]; ls = lab[order]
T = n // 32; TQ = 4
lo = cs[:T * 32].reshape(T, 32, d).min(1)[:, :2]; hi = cs[:T * 32].reshape(T, 32, d).max(1)[:, :2]
tl = ls[:T * 32].reshape(T, 32)[:, 0]
rng = np.random.default_rng(3)
groups = rng.choice(T // TQ, 80, replace=False)
res = {k: 0.0 for k in (1, 2, 3, 5, 9, 17)}; extra = {k: 0.0 for k in res}
for g in groups:
    t0 = g * TQ
    qlo = lo[t0:t0 + TQ].min(0); qhi = hi[t0:t0 + TQ].max(0)
    gg = np.maximum(0, np.maximum(qlo - hi, lo - qhi)); g2 = (gg * gg).sum(1)
    q = cs[t0 * 32:(t0 + TQ) * 32]; fq = fs[t0 * 32:(t0 + TQ) * 32]
    d2 = (q * q).sum(1)[:, None] + (cs * cs).sum(1)[None, :] - 2.0 * q @ cs.T
    d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
    own = ls[None, :] == ls[t0 * 32]
    nn = np.where(own, d2, np.inf).min(1)
    hd = np.where(own & (fs[None, :] < fq[:, None]), d2, np.inf).min(1)
    need = np.maximum(nn, np.where(np.isfinite(hd), hd, 0))      # confirming radius^2 per query
    same = tl == tl[t0]
    srt = np.sort(need)[::-1]
    # per-query point-to-tile gaps for the left-over queries
    for k in res:
        rk = srt[k - 1]
        res[k] += (g2[same] < rk).sum() / T
        worst = np.argsort(need)[::-1][:k - 1]
        for w in worst:
            pg = np.maximum(0, np.maximum(q[w, :2] - hi, lo - q[w, :2])); p2 = (pg * pg).sum(1)
            extra[k] += ((p2[same] < need[w]) & ~(g2[same] < rk)).sum()
for k in res:
    print(f"ring = {k}-th worst query: tile fraction {res[k]/80:.4f} ({res[k]/res[1]:.3f} of now), exact (query, tile) pairs left over per group {extra[k]/80:.0f}")
