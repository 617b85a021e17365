"""How many chains the neighbour sweep needs at C3 if the confirming ring were sized per query TILE instead of per
group of four tiles (the exact nn / nn_hd distances come from the library; orderings and boxes are rebuilt here)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [r])
fe = dens.calculate_free_energies(pops[0].contiguous())
nn_i, nn_d2, hd_i, hd_d2 = dens.nearest_neighbors_partial(c, fe)
print("tile pairs evaluated by the sweep:", dens.evaluated_tiles(c.device)[1], "of", (n // 32) ** 2)
x, y = c[:, 0], c[:, 1]
e0, e1 = (x.max() - x.min()).item(), (y.max() - y.min()).item()
cell = (e0 * e1 * 128.0 / n) ** 0.5
bx = ((x - x.min()) / cell).floor().long(); by = ((y - y.min()) / cell).floor().long()
fq = ((fe - fe.min()) / (fe[fe < 1e30].max() - fe.min())).clamp(0, 1)
key = (bx * (by.max() + 1) + by).double() + fq.double() * 0.999
order = torch.argsort(key)
T = n // 32
co = c[order][:T * 32, :2].reshape(T, 32, 2)
lo, hi = co.min(1).values, co.max(1).values                      # [T, 2]
need = torch.maximum(nn_d2, torch.where(hd_d2 < 1e30, hd_d2, torch.zeros_like(hd_d2)))[order][:T * 32].reshape(T, 32)
need_t = need.max(1).values
need_nn_t = nn_d2[order][:T * 32].reshape(T, 32).max(1).values
feo = fe[order][:T * 32].reshape(T, 32)
fe_lo_t, fe_hi_t = feo.min(1).values, feo.max(1).values
def gap2(qlo, qhi):
    g = torch.clamp(torch.maximum(qlo[None, :] - hi, lo - qhi[None, :]), min=0)
    return (g * g).sum(1)
rng = np.random.default_rng(3)
TQ = 4
tot_g = tot_t = tot_t2 = first = 0.0
nn_part = hd_useful = 0.0
groups = rng.choice(T // TQ, 400, replace=False)
for g in groups:
    t0 = int(g) * TQ
    qlo, qhi = lo[t0:t0 + TQ].min(0).values, hi[t0:t0 + TQ].max(0).values
    g2 = gap2(qlo, qhi)
    diag2 = ((qhi - qlo) ** 2).sum().item()
    r1 = max(diag2, cell * cell)                                  # the first ring of the kernel
    ng = max(need_t[t0:t0 + TQ].max().item(), r1)
    tot_g += TQ * (g2 < ng).sum().item()
    nng = max(need_nn_t[t0:t0 + TQ].max().item(), r1)             # what the plain neighbours of the group need
    qfe_max = fe_hi_t[t0:t0 + TQ].max().item()
    in_ring = g2 < ng
    nn_part += TQ * (g2 < nng).sum().item()
    hd_useful += TQ * (in_ring & (g2 >= nng) & (fe_lo_t < qfe_max)).sum().item()
    first += TQ * (g2 < r1).sum().item()
    for k in range(TQ):
        nk = max(need_t[t0 + k].item(), r1)
        tot_t += (g2 < nk).sum().item()                           # group box, the tile's own need
        tot_t2 += (gap2(lo[t0 + k], hi[t0 + k]) < max(need_t[t0 + k].item(), 0.0)).sum().item()   # tile box, tile need
norm = len(groups) * TQ * T
norm = len(groups) * TQ * T
print(f"of the group ring: tiles the plain neighbours need {nn_part/norm:.4f}; beyond that, tiles that hold a frame of lower free energy than some query {hd_useful/norm:.4f}; skippable {(tot_g-nn_part-hd_useful)/norm:.4f}")
print(f"chains / all tile pairs: first ring {first/norm:.4f}; group ring (now) {tot_g/norm:.4f}; group box + tile need {tot_t/norm:.4f}; tile box + tile need {tot_t2/norm:.4f}")
q = torch.quantile(need_t.float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=need_t.device))
print("need per tile (d2) quantiles 50/90/99/99.9 %:", [round(v, 4) for v in q.tolist()], " cell^2", round(cell * cell, 5))
