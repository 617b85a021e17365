"""Nested equal-count slabs (sort by col a, cut into na slabs; inside each sort by col b, cut into nb; ...)
against the 2-D cells and the k-d order of prune_study2.py: pop(group) / nn(group worst) fractions at C3."""
import numpy as np, sys
sys.path.insert(0, '.')
sys.argv = [sys.argv[0], '1000000', '10', '0.2'] if len(sys.argv) < 4 else sys.argv
exec(open('scratch/prune_study2.py').read().split("mn = c.min(0)")[0])   # n, d, r, c, boxes, gap2, study

def slab_order(dims, counts):
    order = np.arange(n)
    seg = np.zeros(n, dtype=np.int64)           # segment id per position
    for k, m in zip(dims, counts):
        # sort inside segments by column k
        o = np.lexsort((c[order, k], seg))
        order = order[o]; seg = seg[o]
        # cut every segment into m equal-count pieces
        starts = np.flatnonzero(np.r_[True, seg[1:] != seg[:-1]])
        lens = np.diff(np.r_[starts, n])
        pos = np.arange(n) - np.repeat(starts, lens)
        piece = (pos * m) // np.repeat(lens, lens)
        seg = seg * m + piece
    return order

for dims, counts in (((0, 1), (177, 177)), ((0, 1, 2), (32, 32, 32)), ((0, 1, 2), (64, 64, 8)), ((0, 1, 2), (45, 45, 16)),
                     ((0, 1, 2, 3), (16, 16, 16, 8)), ((0, 1, 2, 3), (32, 32, 8, 4))):
    o = slab_order(dims, counts)
    study(f"slabs dims {dims} counts {counts}", o, list(range(len(dims))))
