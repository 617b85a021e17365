"""Independent numpy emulation of the canonical arithmetic.

Used by the tests to cross-check oracle/dc_oracle.c with a second implementation
written differently (vectorised float32 numpy, every op rounds to float32).
"""
import numpy as np


def d2_matrix(c):
    """canonical d2 for all ordered pairs of rows of c (float32 [n, D]) -> float32 [n, n]."""
    c = np.asarray(c, dtype=np.float32)
    n, D = c.shape
    diff = (c[:, None, :] - c[None, :, :]).astype(np.float32)
    p = (diff * diff).astype(np.float32)
    if D <= 3:
        s = p[:, :, 0].copy()
        for k in range(1, D):
            s = (s + p[:, :, k]).astype(np.float32)
        return s
    V = 4 * (D // 4)
    a = [np.zeros((n, n), dtype=np.float32) for _ in range(4)]
    for k0 in range(0, V, 4):
        for l in range(4):
            a[l] = (a[l] + p[:, :, k0 + l]).astype(np.float32)
    s = ((a[0] + a[2]).astype(np.float32) + (a[1] + a[3]).astype(np.float32)).astype(np.float32)
    k = V
    if D - k >= 2:
        s = (s + (p[:, :, k] + p[:, :, k + 1]).astype(np.float32)).astype(np.float32)
        k += 2
    if D - k == 1:
        s = (s + p[:, :, k]).astype(np.float32)
    return s


def populations(c, radii):
    d2 = d2_matrix(c)
    n = d2.shape[0]
    out = np.zeros((len(radii), n), dtype=np.uint64)
    off = ~np.eye(n, dtype=bool)
    for r, rad in enumerate(radii):
        rad = np.float32(rad)
        rad2 = np.float32(rad * rad)
        out[r] = 1 + ((d2 < rad2) & off).sum(axis=1)
    return out


def free_energies(pops):
    pops = np.asarray(pops, dtype=np.uint64)
    max_pop = np.float32(pops.max())
    rec = np.float32(np.float32(1.0) / max_pop)
    q = (pops.astype(np.float32) * rec).astype(np.float32)
    return (-np.log(q.astype(np.float64))).astype(np.float32)


def nearest_neighbors(c, fe):
    d2 = d2_matrix(c)
    n = d2.shape[0]
    fe = np.asarray(fe, dtype=np.float32)
    fmax = np.finfo(np.float32).max
    nn_idx = np.full(n, n + 1, dtype=np.uint64)
    hd_idx = np.full(n, n + 1, dtype=np.uint64)
    nn_d2 = np.full(n, fmax, dtype=np.float32)
    hd_d2 = np.full(n, fmax, dtype=np.float32)
    for i in range(n):
        row = d2[i].copy()
        row[i] = np.inf
        j = int(np.argmin(row))  # first minimum = lowest index
        if n > 1:
            nn_idx[i], nn_d2[i] = j, row[j]
        m = fe < fe[i]
        m[i] = False
        if m.any():
            rowm = np.where(m, row, np.inf)
            j = int(np.argmin(rowm))
            hd_idx[i], hd_d2[i] = j, rowm[j]
    return nn_idx, nn_d2, hd_idx, hd_d2
