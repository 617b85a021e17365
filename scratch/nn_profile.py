"""Rare-path statistics of nn_pruned_kernel (a library built with CXXFLAGS_EXTRA=-DDC_NN_PROFILE: header words 14..19)."""
import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d = 1_000_000, 10
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
dens.nearest_neighbors_partial(c, fe)
torch.cuda.synchronize()
ws = dens._workspace(c.device).buf
h = ws[:1024].view(torch.int64).cpu().tolist()
chains, rare, trig, cand = h[2], h[7], h[8], h[9]
print(f"chains {chains}  rare-path entries {rare} ({rare/chains:.4f} per chain)  parks {trig} ({trig/chains:.4f})  candidates {cand} ({cand/chains:.3f} per chain, {cand/n:.1f} per query)")
