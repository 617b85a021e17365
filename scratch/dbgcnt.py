import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n=1000000
c = torch.from_numpy(gaussian_blobs(n, 10)).cuda()
pops = dens.calculate_populations_partial(c, [0.2], variant='pruned')
fe = dens.calculate_free_energies(pops[0].contiguous())
dens.nearest_neighbors_partial(c, fe, variant='pruned')
torch.cuda.synchronize()
ws = dens._workspace(c.device).buf
hdr = ws[:256].cpu().numpy().view(np.uint64)
# chain_counter for nn = byte 16 -> u64 index 2; +6.. -> idx 8..11
print("nn chains", hdr[2], "trig calls", hdr[8], "special", hdr[9], "rings(sum over waves)", hdr[10], "trig lanes", hdr[11])
print("per chain: trig %.4f special %.4f ; rings/wave %.2f; lanes per trig %.2f" % (hdr[8]/hdr[2], hdr[9]/hdr[2], hdr[10]/7813.0, hdr[11]/max(1,hdr[8])))
