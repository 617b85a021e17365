"""C3 neighbour sweep: does the box area of a query group predict its cost (tiles in its confirming ring)?"""
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
cell = (e0 * e1 * 128.0 / n) ** 0.5
bx = ((x - x.min()) / cell).floor().long(); by = ((y - y.min()) / cell).floor().long()
fq = ((fe - fe.min()) / (fe[fe < 1e30].max() - fe.min())).clamp(0, 1)
order = torch.argsort((bx * (by.max() + 1) + by).double() + fq.double() * 0.999)
TQ = 4; G = n // (32 * TQ); T = G * TQ
co = c[order][:G * TQ * 32, :2]
tl, th = co.reshape(T, 32, 2).min(1).values, co.reshape(T, 32, 2).max(1).values
gl, gh = co.reshape(G, TQ * 32, 2).min(1).values, co.reshape(G, TQ * 32, 2).max(1).values
need = torch.maximum(nn_d2, torch.where(hd_d2 < 1e30, hd_d2, torch.zeros_like(hd_d2)))[order][:G * TQ * 32].reshape(G, TQ * 32).max(1).values
area = ((gh - gl)[:, 0] * (gh - gl)[:, 1])
diag2 = ((gh - gl) ** 2).sum(1)
cost = torch.zeros(G, device=c.device)
for g0 in range(0, G, 256):
    ql, qh = gl[g0:g0 + 256], gh[g0:g0 + 256]                     # [b, 2]
    gp = torch.clamp(torch.maximum(ql[:, None, :] - th[None], tl[None] - qh[:, None, :]), min=0)
    g2 = (gp * gp).sum(2)                                         # [b, T]
    ng = torch.maximum(need[g0:g0 + 256], torch.clamp(diag2[g0:g0 + 256], min=cell * cell))
    cost[g0:g0 + 256] = (g2 < ng[:, None]).sum(1).float()
cs, idx = torch.sort(cost, descending=True)
print("tiles in the ring per group: mean %.0f  median %.0f  p99 %.0f  max %.0f" % (cost.mean(), cost.median(), torch.quantile(cost, 0.99), cost.max()))
for frac in (0.01, 0.03, 0.1):
    k = int(G * frac)
    top_area = torch.topk(area, k).indices
    top_cost = idx[:k]
    hit = len(set(top_area.tolist()) & set(top_cost.tolist())) / k
    print(f"top {frac:.0%} by box area: mean cost {cost[top_area].mean():.0f} (all: {cost.mean():.0f}); overlap with the top {frac:.0%} by cost {hit:.2f}")
print("correlation(cost, area) %.3f   correlation(cost, need) %.3f" % (torch.corrcoef(torch.stack([cost, area]))[0, 1], torch.corrcoef(torch.stack([cost, need]))[0, 1]))
ra = area / (cell * cell)
print("group box area / cell^2 quantiles 50/90/97/99/99.9 %:", [round(v, 2) for v in torch.quantile(ra, torch.tensor([0.5, 0.9, 0.97, 0.99, 0.999], device=ra.device)).tolist()])
for k in (1.0, 2.0, 4.0, 8.0):
    sel = ra > k
    print(f"area > {k} cell^2: {sel.float().mean().item():.4f} of the groups, mean cost {cost[sel].mean().item():.0f}, they hold {(cost[sel].sum() / cost.sum()).item():.3f} of the work; heaviest 0.5 % by cost inside: {(sel[idx[:G // 200]]).float().mean().item():.2f}")
