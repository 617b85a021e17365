"""Neighbour sweep: query groups of one cell (four free-energy quartiles of 128 frames: now) against groups of the SAME
quartile of 2 x 2 neighbouring cells (homogeneous confirming radii, a box twice as wide).  Tile pairs under the kernel's
rule (group box gap^2 < the group's worst confirming distance), C3 data, density-adapted cells."""
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

def order(mode):
    keys = np.zeros(n, dtype=np.int64)
    for k in range(3):
        m = lab == k; x = c[m]; lo = x[:, :2].min(0)
        edge = np.sqrt(np.pi * (2.5 * sig) ** 2 / (m.sum() / 128))
        bx = ((x[:, 0] - lo[0]) / edge).astype(np.int64); by = ((x[:, 1] - lo[1]) / edge).astype(np.int64)
        f = fe[m]
        if mode == 'cell':
            fq = ((f - f.min()) / (f.max() - f.min()) * 255).astype(np.int64)
            keys[m] = ((k * 4096 + bx) * 4096 + by) * 256 + fq
        elif mode.startswith('bin') or mode == 'blockfe':
            fr = (f - fe.min()) / (fe.max() - fe.min())
            blk = (bx // 2) * 4096 + (by // 2); sub = (bx % 2) * 2 + (by % 2)
            if mode == 'blockfe':
                keys[m] = (k * (1 << 24) + blk) * 4096 + (fr * 4095).astype(np.int64)
            else:
                nb = int(mode[3:])
                keys[m] = ((((k * (1 << 24) + blk) * nb + np.minimum((fr * nb).astype(np.int64), nb - 1)) * 4 + sub) * 64) + ((fr * nb * 64).astype(np.int64) % 64)
        else:
            # quartile of the frame inside its own cell (rank-based), then (block, quartile, cell-in-block, fe)
            cell = bx * 4096 + by
            o = np.lexsort((f, cell)); rank = np.empty(len(f), dtype=np.int64)
            cs_ = cell[o]; starts = np.flatnonzero(np.r_[True, cs_[1:] != cs_[:-1]]); lens = np.diff(np.r_[starts, len(f)])
            pos = np.arange(len(f)) - np.repeat(starts, lens)
            q = np.minimum(pos * 4 // np.repeat(lens, lens), 3)
            quart = np.empty(len(f), dtype=np.int64); quart[o] = q
            fq = ((f - f.min()) / (f.max() - f.min()) * 63).astype(np.int64)
            blk = (bx // 2) * 4096 + (by // 2); sub = (bx % 2) * 2 + (by % 2)
            keys[m] = ((((k * (1 << 24) + blk) * 4 + quart) * 4 + sub) * 64) + fq
    return np.argsort(keys, kind='stable')

def study(name, o, nq=120, TQ=4):
    cs = c[o]; fs = fe[o]; ls = lab[o]
    T = n // 32
    lo = cs[:T * 32].reshape(T, 32, d).min(1)[:, :2]; hi = cs[:T * 32].reshape(T, 32, d).max(1)[:, :2]
    tl = ls[:T * 32].reshape(T, 32)[:, 0]
    rng = np.random.default_rng(9)
    tot = 0.0
    for g in rng.choice(T // TQ, nq, replace=False):
        t0 = g * TQ
        qlo = lo[t0:t0 + TQ].min(0); qhi = hi[t0:t0 + TQ].max(0)
        gg = np.maximum(0, np.maximum(qlo - hi, lo - qhi)); g2 = (gg * gg).sum(1)
        q = cs[t0 * 32:(t0 + TQ) * 32]; fq = fs[t0 * 32:(t0 + TQ) * 32]
        d2 = (q * q).sum(1)[:, None] + (cs * cs).sum(1)[None, :] - 2.0 * q @ cs.T
        d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
        own = ls[None, :] == ls[t0 * 32]
        nn = np.where(own, d2, np.inf).min(1)
        hd = np.where(own & (fs[None, :] < fq[:, None]), d2, np.inf).min(1)
        need = np.maximum(nn, np.where(np.isfinite(hd), hd, 0)).max()
        tot += (g2[tl == tl[t0]] < need).sum() / T
    print(f"{name:50s} tile fraction {tot/nq:.4f}")

study("one cell per group (now)", order('cell'))
study("same quartile of 2 x 2 cells per group", order('quart'))
study("2 x 2 block, free energy only", order('blockfe'))
for nb in (8, 16, 32):
    study(f"2 x 2 block, {nb} global FE bins, sub-cell, FE", order(f'bin{nb}'))
