import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(5000000, 30)).cuda()
radii = [0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 0.65]
dens.calculate_populations_segment(c, radii, 3, 8)
torch.cuda.synchronize()
a, b = dens.evaluated_tiles(c.device)
print("chains", a, "k>=1", b & 0xFFFFFFFF, "k>=2", b >> 32)
