"""Neighbour sweep: tile pairs a query group of four tiles must evaluate (box gap^2 below the group's worst
confirming distance, nn AND nn with lower free energy) under the current 2-D cell order (cells of ~64 frames, free
energy inside) and under nested equal-count slabs on columns 0/1 with column 2 inside (3-D boxes).  Free energies: the
analytic mixture density (a stand-in for the populations)."""
import numpy as np, sys
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = gaussian_blobs(n, d)
cent = np.array([(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)], dtype=np.float32)
sig = 0.08
# density proxy in 10-D: blob k centred at (cent_k, 0...) 
def dens(x):
    out = np.zeros(len(x))
    for k in range(3):
        mu = np.zeros(d, dtype=np.float32); mu[:2] = cent[k]
        out += np.exp(-((x - mu) ** 2).sum(1) / (2 * sig * sig))
    return out
fe = -np.log(dens(c) + 1e-300)
lab = np.argmin(((c[:, None, :2] - cent[None]) ** 2).sum(2), 1)

def order_2d(cell_frames=64):
    keys = np.zeros(n, dtype=np.int64)
    for k in range(3):
        m = lab == k
        x = c[m]
        lo = x[:, :2].min(0); hi = x[:, :2].max(0)
        # cell edge so that occupied area / cells ~ cell_frames per cell (rough: use 6 sigma square)
        ncell = m.sum() / cell_frames
        edge = np.sqrt(np.pi * (2.5 * sig) ** 2 / ncell)
        bx = ((x[:, 0] - lo[0]) / edge).astype(np.int64); by = ((x[:, 1] - lo[1]) / edge).astype(np.int64)
        f = fe[m]; fq = ((f - f.min()) / (f.max() - f.min()) * 255).astype(np.int64)
        keys[m] = ((k * 4096 + bx) * 4096 + by) * 256 + fq
    return np.argsort(keys, kind='stable')

def order_slab(s=None):
    order = []
    for k in range(3):
        idx = np.flatnonzero(lab == k)
        nc = len(idx)
        sc = s or int(round((nc / 32) ** (1 / 3)))
        o = idx[np.argsort(c[idx, 0], kind='stable')]
        for a in np.array_split(o, sc):
            a = a[np.argsort(c[a, 1], kind='stable')]
            for b in np.array_split(a, sc):
                order.append(b[np.argsort(c[b, 2], kind='stable')])
    return np.concatenate(order)

def study(name, order, dims, nq=80, TQ=4):
    cs = c[order]; fs = fe[order]; ls = lab[order]
    T = n // 32
    lo = cs[:T * 32].reshape(T, 32, d).min(1); hi = cs[:T * 32].reshape(T, 32, d).max(1)
    tl = ls[:T * 32].reshape(T, 32)[:, 0]
    rng = np.random.default_rng(1)
    groups = rng.choice(T // TQ, nq, replace=False)
    f_nn = f_hd = 0.0
    for g in groups:
        t0 = g * TQ
        qlo = lo[t0:t0 + TQ].min(0); qhi = hi[t0:t0 + TQ].max(0)
        gg = np.maximum(0, np.maximum(qlo[dims] - hi[:, dims], lo[:, dims] - qhi[dims]))
        g2 = (gg * gg).sum(1)
        q = cs[t0 * 32:(t0 + TQ) * 32]; fq = fs[t0 * 32:(t0 + TQ) * 32]
        d2 = (q * q).sum(1)[:, None] + (cs * cs).sum(1)[None, :] - 2.0 * q @ cs.T
        d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
        nn = d2.min(1)
        d2h = np.where(fs[None, :] < fq[:, None], d2, np.inf)
        hd = d2h.min(1)
        same = tl == tl[t0]          # (own component only: the rest is the cross search)
        hd_own = np.where(np.isfinite(hd), hd, 0)
        # confirming radius of the group (own component): worst of nn and hd (hd capped to the component)
        d2h_own = np.where((fs[None, :] < fq[:, None]) & (ls[None, :] == ls[t0 * 32]), d2, np.inf).min(1)
        worst_hd = np.where(np.isfinite(d2h_own), d2h_own, 0).max()
        f_nn += (g2[same] < max(nn.max(), 1e-12)).sum() / T
        f_hd += (g2[same] < max(nn.max(), worst_hd)).sum() / T
    print(f"{name:40s} nn only {f_nn/nq:.4f}   nn+hd {f_hd/nq:.4f}")

study("2-D cells (64 frames, FE inside)", order_2d(64), [0, 1])
study("2-D cells (32 frames, FE inside)", order_2d(32), [0, 1])
study("slabs s=cbrt(n/32), col2 inside, 3-D", order_slab(), [0, 1, 2])
study("slabs s=16, 3-D boxes", order_slab(16), [0, 1, 2])
study("slabs s=28, 3-D boxes", order_slab(28), [0, 1, 2])
study("slabs s=cbrt, 2-D boxes only", order_slab(), [0, 1])
