"""VERDICT r5 item 5, CPU-costed first: a first-level structure in ALL columns for the NEIGHBOUR sweep on data without
structure -- cells = the Voronoi regions of K k-means centres in the full space (K = n / 128 as the 2-D cells of the
current order; MiniBatchKMeans fitted on a 100 000-row sample, every row assigned to its nearest centre), frames of a cell
by free energy, tiles of 32 consecutive frames, and per tile BOTH bounds the full-space structure offers: the box in all
D columns and the ball (centroid, radius).  For sampled query groups (6 tiles) the tile pairs the sweep must evaluate =
reference tiles whose lower bound of the squared distance to the group (max of box gap^2 and ball gap^2) is below the
group's confirming radius^2 (largest nn / lower-free-energy nn distance of its queries, exact, brute force) -- the kernel's
own ring rule, as scratch/ring_cols_study.py -- against the current order (2-D cells on columns 0/1, 2-D boxes) on
  the uniform box (1M x 10 in [0, 1]^10: profiles/r5_unfav_uniform.json evaluates 70 % of N^2),
  one broad blob (1M x 10, sigma 0.08: 52 %) and C3's three blobs (20 %).
Go only if the uniform box drops below 45 % of N^2 without C3 rising above 21 %."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
from sklearn.cluster import MiniBatchKMeans

n, d, TQ = 1_000_000, 10, 6
rng = np.random.default_rng(11)


def order_2d(c, fe, frames_per_cell=128.0):
    k = 2
    lo, hi = c[:, :k].min(0), c[:, :k].max(0)
    ext = np.maximum(hi - lo, 1e-9)
    edge = (np.prod(ext) * frames_per_cell / len(c)) ** (1.0 / k)
    idx = np.minimum(((c[:, :k] - lo) / edge).astype(np.int64), 4000)
    nb = idx.max(0) + 1
    cell = idx[:, 0] * nb[1] + idx[:, 1]
    fq = ((fe - fe.min()) / max(fe.max() - fe.min(), 1e-30) * 511).astype(np.int64)
    return np.argsort(cell * 512 + fq, kind='stable')


def order_kmeans(c, fe, frames_per_cell=128.0):
    K = max(8, int(len(c) / frames_per_cell))
    t0 = time.time()
    km = MiniBatchKMeans(n_clusters=K, batch_size=20000, n_init=1, max_iter=20, random_state=3)
    km.fit(c[rng.choice(len(c), 100_000, replace=False)])
    cen = km.cluster_centers_.astype(np.float32)
    c2 = (cen * cen).sum(1)
    cell = np.empty(len(c), dtype=np.int64)
    for s in range(0, len(c), 20000):
        x = c[s:s + 20000]
        cell[s:s + 20000] = (c2[None, :] - 2.0 * (x @ cen.T)).argmin(1)
    # cells numbered along the first principal direction of the centres (neighbouring cells near each other in the order)
    u = np.linalg.svd(cen - cen.mean(0), full_matrices=False)[2][0]
    rank = np.argsort(np.argsort(cen @ u))
    fq = ((fe - fe.min()) / max(fe.max() - fe.min(), 1e-30) * 511).astype(np.int64)
    print(f"   k-means cells: K = {K}, {time.time() - t0:.0f} s, frames per cell {np.bincount(cell, minlength=K).mean():.0f} "
          f"(max {np.bincount(cell, minlength=K).max()})", flush=True)
    return np.argsort(rank[cell] * 512 + fq, kind='stable')


def study(name, c, fe, order, kcols, ball, groups=24):
    cs, fs = c[order], fe[order]
    T = len(c) // 32
    tiles = cs[:T * 32].reshape(T, 32, d)
    lo, hi = tiles[:, :, :kcols].min(1), tiles[:, :, :kcols].max(1)
    cen = tiles.mean(1)
    rad = np.sqrt(((tiles - cen[:, None, :]) ** 2).sum(2).max(1))
    sq = (cs * cs).sum(1)
    frac = []
    for g in rng.choice(T // TQ, groups, replace=False):
        t0 = g * TQ
        q, fq = cs[t0 * 32:(t0 + TQ) * 32], fs[t0 * 32:(t0 + TQ) * 32]
        d2 = (q * q).sum(1)[:, None] + sq[None, :] - 2.0 * (q @ cs.T)
        d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
        nn = d2.min(1)
        hd = np.where(fs[None, :] < fq[:, None], d2, np.inf).min(1)
        hd = np.where(np.isfinite(hd), hd, 0.0)
        confirm = max(nn.max(), hd.max())
        qlo, qhi = lo[t0:t0 + TQ].min(0), hi[t0:t0 + TQ].max(0)
        gap = np.maximum(0.0, np.maximum(qlo - hi, lo - qhi))
        bound = (gap * gap).sum(1)
        if ball:
            qc = q.mean(0)
            qr = np.sqrt(((q - qc) ** 2).sum(1).max())
            bg = np.maximum(0.0, np.sqrt(((cen - qc) ** 2).sum(1)) - rad - qr)
            bound = np.maximum(bound, bg * bg)
        frac.append(float((bound < confirm).mean()))
    print(f"{name:34s} evaluated tile pairs {np.mean(frac):.3f} of all (min {np.min(frac):.3f}, max {np.max(frac):.3f})", flush=True)
    return float(np.mean(frac))


def fe_of_blobs(c, cents):
    dens = np.zeros(len(c))
    for cen in cents:
        mu = np.zeros(d, dtype=np.float32)
        mu[:len(cen)] = cen
        dens += np.exp(-((c - mu) ** 2).sum(1) / (2 * 0.08 ** 2))
    return -np.log(dens + 1e-300)


sets = {}
uni = rng.random((n, d), dtype=np.float32)
sets['uniform box'] = (uni, -np.log((1 + rng.poisson(0.16, n)) / 8.0))
blob = (rng.standard_normal((n, d)) * 0.08).astype(np.float32)
sets['one broad blob'] = (blob, fe_of_blobs(blob, [()]))
c3 = gaussian_blobs(n, d)
sets['C3 blobs'] = (c3, fe_of_blobs(c3, [(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)]))
res = {}
for name, (c, fe) in sets.items():
    res[(name, '2d')] = study(f"{name}: 2-D cells, 2-D boxes", c, fe, order_2d(c, fe), 2, False)
    ok = order_kmeans(c, fe)
    res[(name, 'km_box')] = study(f"{name}: k-means cells, 10-D boxes", c, fe, ok, d, False)
    res[(name, 'km')] = study(f"{name}: k-means cells, boxes + balls", c, fe, ok, d, True)
go = res[('uniform box', 'km')] < 0.45 and res[('C3 blobs', 'km')] <= 0.21
print(f"uniform {res[('uniform box', 'km')]:.3f} (< 0.45 ?)  C3 {res[('C3 blobs', 'km')]:.3f} (<= 0.21 ?)  one blob {res[('one broad blob', 'km')]:.3f}  -> {'GO' if go else 'no go'}")
