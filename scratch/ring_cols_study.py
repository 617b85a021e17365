"""VERDICT r4 item 6, CPU-costed first: would ring bounds from 3 - 4 columns help the NEIGHBOUR sweep on data without
structure?  For sampled query groups (6 tiles = 192 consecutive frames of the sweep's order) the tile pairs the sweep must
evaluate = reference tiles whose box gap^2 (in the columns the boxes are kept in) is below the group's confirming radius^2
(the largest nn / lower-free-energy nn distance of its queries, exact, brute force against all frames) -- under
  k = 2   the current order: cells of ~128 frames on columns 0/1, free energy inside, 2-D boxes
  k = 3   cells on columns 0..2, 3-D boxes
  k = 4   cells on columns 0..3, 4-D boxes
on the uniform box (1M x 10 in [0, 1]^10: profiles/r4_unfav_uniform.json evaluates 70 % of N^2) and on C3's blobs.
Go only if the uniform box drops below 45 % without C3 rising above 21 %.
Free energies: uniform box -- populations at r = 0.2 are 1 + Poisson(0.16) there (mean 1.16, unfav bench), drawn
independently; C3 -- the analytic mixture density (as scratch/nn3d_study.py)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs

n, d, TQ = 1_000_000, 10, 6
rng = np.random.default_rng(7)


def cells_order(c, fe, k, frames_per_cell=128.0):
    lo, hi = c[:, :k].min(0), c[:, :k].max(0)
    ext = np.maximum(hi - lo, 1e-9)
    edge = (np.prod(ext) * frames_per_cell / len(c)) ** (1.0 / k)
    idx = np.minimum(((c[:, :k] - lo) / edge).astype(np.int64), 4000)
    nb = idx.max(0) + 1
    cell = np.zeros(len(c), dtype=np.int64)
    for j in range(k):
        cell = cell * nb[j] + idx[:, j]
    fq = ((fe - fe.min()) / max(fe.max() - fe.min(), 1e-30) * 511).astype(np.int64)
    return np.argsort(cell * 512 + fq, kind='stable')


def study(name, c, fe, k, groups=24):
    order = cells_order(c, fe, k)
    cs, fs = c[order], fe[order]
    T = len(c) // 32
    lo = cs[:T * 32].reshape(T, 32, d)[:, :, :k].min(1)
    hi = cs[:T * 32].reshape(T, 32, d)[:, :, :k].max(1)
    sq = (cs * cs).sum(1)
    frac = []
    for g in rng.choice(T // TQ, groups, replace=False):
        t0 = g * TQ
        q, fq = cs[t0 * 32:(t0 + TQ) * 32], fs[t0 * 32:(t0 + TQ) * 32]
        d2 = (q * q).sum(1)[:, None] + sq[None, :] - 2.0 * (q @ cs.T)
        d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
        nn = d2.min(1)
        hd = np.where(fs[None, :] < fq[:, None], d2, np.inf).min(1)
        hd = np.where(np.isfinite(hd), hd, 0.0)          # (the free-energy minimum: nothing to confirm)
        confirm = max(nn.max(), hd.max())
        qlo, qhi = lo[t0:t0 + TQ].min(0), hi[t0:t0 + TQ].max(0)
        gap = np.maximum(0.0, np.maximum(qlo - hi, lo - qhi))
        frac.append(float(((gap * gap).sum(1) < confirm).mean()))
    print(f"{name:28s} k = {k}: evaluated tile pairs {np.mean(frac):.3f} of all (min {np.min(frac):.3f}, max {np.max(frac):.3f})", flush=True)
    return float(np.mean(frac))


uni = rng.random((n, d), dtype=np.float32)
fe_uni = -np.log((1 + rng.poisson(0.16, n)) / 8.0)
c3 = gaussian_blobs(n, d)
cent = np.array([(-1.0, -0.5), (0.0, 0.5), (1.0, -0.5)], dtype=np.float32)
dens = np.zeros(n)
for kk in range(3):
    mu = np.zeros(d, dtype=np.float32)
    mu[:2] = cent[kk]
    dens += np.exp(-((c3 - mu) ** 2).sum(1) / (2 * 0.08 ** 2))
fe_c3 = -np.log(dens + 1e-300)
res = {}
for k in (2, 3, 4):
    res[('uniform', k)] = study("uniform box 1M x 10", uni, fe_uni, k)
    res[('c3', k)] = study("C3 blobs 1M x 10", c3, fe_c3, k)
for k in (3, 4):
    go = res[('uniform', k)] < 0.45 and res[('c3', k)] <= 0.21
    print(f"k = {k}: uniform {res[('uniform', k)]:.3f} (< 0.45 ?)  C3 {res[('c3', k)]:.3f} (<= 0.21 ?)  -> {'GO' if go else 'no go'}")
