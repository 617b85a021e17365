"""C3: of the reference tiles a query group's 2-D box test lets through, how many could a full-dimensional
centroid + radius test still reject?  (populations at r = 0.2; neighbours at the group's worst need)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, r = 1_000_000, 10, 0.2
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [r])
fe = dens.calculate_free_energies(pops[0].contiguous())
nn_i, nn_d2, hd_i, hd_d2 = dens.nearest_neighbors_partial(c, fe)
x, y = c[:, 0], c[:, 1]
e0, e1 = (x.max() - x.min()).item(), (y.max() - y.min()).item()
def study(frames_per_cell, TQ, use_need):
    cell = (e0 * e1 * frames_per_cell / n) ** 0.5
    bx = ((x - x.min()) / cell).floor().long(); by = ((y - y.min()) / cell).floor().long()
    order = torch.argsort((bx * (by.max() + 1) + by).double(), stable=True)
    G = n // (32 * TQ); T = G * TQ
    co = c[order][:T * 32]
    t2 = co[:, :2].reshape(T, 32, 2); tl, th = t2.min(1).values, t2.max(1).values
    tc = co.reshape(T, 32, d).mean(1); trad = (co.reshape(T, 32, d) - tc[:, None, :]).norm(dim=2).max(1).values
    g2 = co[:, :2].reshape(G, TQ * 32, 2); gl, gh = g2.min(1).values, g2.max(1).values
    gc = co.reshape(G, TQ * 32, d).mean(1); grad = (co.reshape(G, TQ * 32, d) - gc[:, None, :]).norm(dim=2).max(1).values
    need = torch.maximum(nn_d2, torch.where(hd_d2 < 1e30, hd_d2, torch.zeros_like(hd_d2)))[order][:T * 32].reshape(G, TQ * 32).max(1).values
    rng = np.random.default_rng(2)
    surv = rej = 0
    for g in rng.choice(G, 400, replace=False):
        g = int(g)
        gp = torch.clamp(torch.maximum(gl[g][None] - th, tl - gh[g][None]), min=0)
        rr2 = need[g].item() if use_need else r * r
        ok = (gp * gp).sum(1) < rr2
        lower = (tc - gc[g][None]).norm(dim=1) - trad - grad[g]
        out = ok & (lower > 0) & (lower * lower >= rr2)
        surv += ok.sum().item(); rej += out.sum().item()
    return surv / (400 * T), rej / max(surv, 1)
print("populations (cells of 64, groups of 6): survivors %.3f of the tiles, of which a centroid + radius test rejects %.3f" % study(64.0, 6, False))
print("neighbours (cells of 128, groups of 4): survivors %.3f of the tiles, of which a centroid + radius test rejects %.3f" % study(128.0, 4, True))

# the same with the frames of a cell sorted by blob (labels from three far-apart seeds, one assignment pass)
seeds = [0]
for _ in range(2):
    dmin = torch.stack([(c - c[s]).norm(dim=1) for s in seeds]).min(0).values
    seeds.append(int(dmin.argmax()))
lab = torch.stack([(c - c[s]).norm(dim=1) for s in seeds]).argmin(0)
print("blob sizes", [int((lab == k).sum()) for k in range(3)])
def study2(frames_per_cell, TQ, use_need):
    cell = (e0 * e1 * frames_per_cell / n) ** 0.5
    bx = ((x - x.min()) / cell).floor().long(); by = ((y - y.min()) / cell).floor().long()
    order = torch.argsort(((bx * (by.max() + 1) + by) * 4 + lab).double(), stable=True)
    G = n // (32 * TQ); T = G * TQ
    co = c[order][:T * 32]
    t2 = co[:, :2].reshape(T, 32, 2); tl, th = t2.min(1).values, t2.max(1).values
    tc = co.reshape(T, 32, d).mean(1); trad = (co.reshape(T, 32, d) - tc[:, None, :]).norm(dim=2).max(1).values
    g2 = co[:, :2].reshape(G, TQ * 32, 2); gl, gh = g2.min(1).values, g2.max(1).values
    gc = co.reshape(G, TQ * 32, d).mean(1); grad = (co.reshape(G, TQ * 32, d) - gc[:, None, :]).norm(dim=2).max(1).values
    need = torch.maximum(nn_d2, torch.where(hd_d2 < 1e30, hd_d2, torch.zeros_like(hd_d2)))[order][:T * 32].reshape(G, TQ * 32).max(1).values
    rng = np.random.default_rng(2)
    surv = rej = 0
    for g in rng.choice(G, 400, replace=False):
        g = int(g)
        gp = torch.clamp(torch.maximum(gl[g][None] - th, tl - gh[g][None]), min=0)
        rr2 = need[g].item() if use_need else r * r
        ok = (gp * gp).sum(1) < rr2
        lower = (tc - gc[g][None]).norm(dim=1) - trad - grad[g]
        out = ok & (lower > 0) & (lower * lower >= rr2)
        surv += ok.sum().item(); rej += out.sum().item()
    return surv / (400 * T), rej / max(surv, 1)
print("cells sorted by blob -- populations: survivors %.3f, rejected by centroid + radius %.3f" % study2(64.0, 6, False))
print("cells sorted by blob -- neighbours:  survivors %.3f, rejected by centroid + radius %.3f" % study2(128.0, 4, True))
