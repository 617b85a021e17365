"""Pruning potential of multi-dimensional orderings at C3 (numpy study, CPU).
k-d ordering (median split on the widest of the first `ds` columns, leaves of 32 frames) against the
current 2-D cell order; populations at r (box gap < r) and neighbours (box gap < the worst nn distance
of the query tile / of a 4-tile query group)."""
import numpy as np, sys
sys.path.insert(0, '.')
from clustering_amd.synth import gaussian_blobs
n, d, r = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
c = gaussian_blobs(n, d)

def kd_order(idx, dims, leaf=32):
    out = []
    stack = [idx]
    while stack:
        ix = stack.pop()
        if len(ix) <= leaf:
            out.append(ix); continue
        sub = c[ix][:, dims]
        k = dims[int(np.argmax(sub.max(0) - sub.min(0)))]
        # split at a multiple of `leaf` so that leaves are full tiles
        half = (len(ix) // leaf // 2) * leaf
        if half == 0: half = leaf
        part = np.argpartition(c[ix, k], half - 1 if half < len(ix) else len(ix) - 1)
        stack.append(ix[part[half:]]); stack.append(ix[part[:half]])
    return np.concatenate(out)

def boxes(order, G=1):
    T = n // (32 * G)
    cs = c[order][:T * 32 * G].reshape(T, 32 * G, d)
    return cs.min(1), cs.max(1)

def gap2(lo, hi, qlo, qhi, dims):
    g = np.maximum(0, np.maximum(qlo[dims] - hi[:, dims], lo[:, dims] - qhi[dims]))
    return (g * g).sum(1)

def study(name, order, dims, nq=60, TQ=4):
    lo, hi = boxes(order)
    T = lo.shape[0]
    rng = np.random.default_rng(1)
    groups = rng.choice(T // TQ, nq, replace=False)
    cs = c[order]
    fp = fn = fn1 = 0.0
    for g in groups:
        t0 = g * TQ
        qlo = lo[t0:t0 + TQ].min(0); qhi = hi[t0:t0 + TQ].max(0)
        g2 = gap2(lo, hi, qlo, qhi, dims)
        fp += (g2 < r * r).mean()
        # true nn distance of the group's queries (brute force)
        q = cs[t0 * 32:(t0 + TQ) * 32]
        d2 = (q * q).sum(1)[:, None] + (cs * cs).sum(1)[None, :] - 2.0 * q @ cs.T
        d2[np.arange(len(q)), np.arange(t0 * 32, (t0 + TQ) * 32)] = np.inf
        nn = np.sqrt(np.maximum(d2.min(1), 0))
        fn += (g2 < nn.max() ** 2).mean()            # the ring the worst query of the group needs
        fn1 += np.mean([(gap2(lo, hi, lo[t0 + k], hi[t0 + k], dims) < nn[32 * k:32 * k + 32].max() ** 2).mean() for k in range(TQ)])
    print(f"{name:44s} pop(group) {fp/nq:.3f}  nn(group worst) {fn/nq:.4f}  nn(tile worst) {fn1/nq:.4f}")

mn = c.min(0)
cell = 0.02
key = np.floor((c[:, 0] - mn[0]) / cell).astype(np.int64) * 100000 + np.floor((c[:, 1] - mn[1]) / cell).astype(np.int64)
o = np.argsort(key, kind='stable')
study("2-D cells (0.02), 2-D boxes", o, [0, 1])
for ds in (2, 3, 4, 6, d):
    o = kd_order(np.arange(n), list(range(ds)))
    study(f"k-d on first {ds} cols, D-dim boxes", o, list(range(d)))
    if ds < d: study(f"k-d on first {ds} cols, {ds}-dim boxes", o, list(range(ds)))
