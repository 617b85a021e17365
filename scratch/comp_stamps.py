"""phases of components_kernel (`git apply -p0 scratch/r6_wave_stamps.patch`, then a build with -DDC_COMP_STAMPS: wall_clock64 stamps in words 8.. of the component region)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
c = torch.from_numpy(gaussian_blobs(n, 10)).cuda()
for it in range(3):
    dens.calculate_populations_partial(c, [0.2])
    torch.cuda.synchronize()
    ws = dens._workspace(c.device).buf
    w = ws[1024:1024 + 64 * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    st = w[8:15]
    names = ["stats_reduce", "cell scan", "labelling", "roots/ids", "origins/adj", "fine grid"]
    print("iter", w[23], " ".join("%s %.1f us" % (nm, ((st[k + 1] - st[k]) & 0xFFFFFFFF) / 100.0) for k, nm in enumerate(names)), "total %.1f" % (((st[6] - st[0]) & 0xFFFFFFFF) / 100.0))
