"""Rare-path statistics and cycle split of nn_pruned_kernel (a library built with CXXFLAGS_EXTRA=-DDC_NN_PROFILE: header
words 14..19 and 32..39), unsharded and as one segment of eight."""
import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
for G in (1, 8):
    pops = dens.calculate_populations_partial(c, [0.2])          # (rebuilds the header: counters start at zero)
    fe = dens.calculate_free_energies(pops[0].contiguous())
    if G == 1:
        dens.nearest_neighbors_partial(c, fe, stats_valid=True)
    else:
        dens.nearest_neighbors_segment(c, fe, 3, G, stats_valid=True)
    torch.cuda.synchronize()
    h = dens._workspace(c.device).buf[:1024].view(torch.int64).cpu().tolist()
    chains, rare, trig, cand = h[2], h[7], h[8], h[9]
    waves, cyc, setup, first, ta, tb = h[16], h[17], h[18], h[19], h[20], h[21]
    print(f"G={G}: chains {chains}  rare-path entries {rare} ({rare/chains:.4f} per chain)  parks {trig}  candidates {cand}")
    print(f"      waves {waves}  cycles per wave {cyc/waves:.0f}  tiles' loads done {ta/waves:.0f}  rows in LDS {tb/waves:.0f}  set-up done (seeds) {setup/waves:.0f} ({setup/cyc:.3f})  first chain {first/waves:.0f} ({first/cyc:.3f})")
