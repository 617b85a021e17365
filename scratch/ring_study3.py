"""Neighbour sweep at C3: chains needed with 2-D cells + free-energy order (now) against 2-D cells x slabs of column 2
x free-energy order with 3-D boxes (exact nn / nn_hd distances from the library)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [r])
fe = dens.calculate_free_energies(pops[0].contiguous())
nn_i, nn_d2, hd_i, hd_d2 = dens.nearest_neighbors_partial(c, fe)
x, y, z = c[:, 0], c[:, 1], c[:, 2]
e0, e1 = (x.max() - x.min()).item(), (y.max() - y.min()).item()
fq = ((fe - fe.min()) / (fe[fe < 1e30].max() - fe.min())).clamp(0, 1).double() * 0.999
need_f = torch.maximum(nn_d2, torch.where(hd_d2 < 1e30, hd_d2, torch.zeros_like(hd_d2)))
T = n // 32
def study(name, frames_per_cell, slab_w, dims):
    cell = (e0 * e1 * frames_per_cell / n) ** 0.5
    bx = ((x - x.min()) / cell).floor().long(); by = ((y - y.min()) / cell).floor().long()
    key = (bx * (by.max() + 1) + by).double()
    if slab_w:
        bz = ((z - z.min()) / slab_w).floor().long()
        key = key * (bz.max() + 1).double() + bz.double()
    order = torch.argsort(key + fq)
    co = c[order][:T * 32, :3].reshape(T, 32, 3)
    lo, hi = co.min(1).values[:, :dims], co.max(1).values[:, :dims]
    need_t = need_f[order][:T * 32].reshape(T, 32).max(1).values
    feo = fe[order][:T * 32].reshape(T, 32)
    mixed = 0.0
    rng = np.random.default_rng(3)
    TQ = 4
    tot = 0.0
    groups = rng.choice(T // TQ, 300, replace=False)
    for g in groups:
        t0 = int(g) * TQ
        qlo, qhi = lo[t0:t0 + TQ].min(0).values, hi[t0:t0 + TQ].max(0).values
        gp = torch.clamp(torch.maximum(qlo[None, :] - hi, lo - qhi[None, :]), min=0)
        g2 = (gp * gp).sum(1)
        r1 = max(((qhi[:2] - qlo[:2]) ** 2).sum().item(), cell * cell)
        ng = max(need_t[t0:t0 + TQ].max().item(), r1)
        sel = g2 < ng
        tot += TQ * sel.sum().item()
        # tiles in the ring whose free-energy range straddles the median query of the group
        qf = feo[t0:t0 + TQ].median().item()
        mixed += ((feo[sel].min(1).values < qf) & (feo[sel].max(1).values >= qf)).float().mean().item()
    print(f"{name:50s} chains / all tile pairs {tot / (len(groups) * TQ * T):.4f}   mixed-FE tiles in the ring {mixed / len(groups):.3f}")
study("2-D cells of ~128 frames, FE order (now)", 128.0, 0.0, 2)
for fpc, w in ((128.0, 0.1), (256.0, 0.1), (256.0, 0.05), (512.0, 0.05), (512.0, 0.03), (1024.0, 0.03)):
    study(f"cells of ~{fpc:.0f} frames x slabs of {w} in column 2, FE", fpc, w, 3)
