import sys, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
c = torch.from_numpy(gaussian_blobs(n, 10)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for _ in range(8):
    ev[0].record()
    p = dens.calculate_populations_partial(c, [0.1, 0.2, 0.3])
    fe = dens.calculate_free_energies(p[1].contiguous())
    ev[1].record(); torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("C2-like step n=%d: min %.3f ms median %.3f" % (n, min(ts), sorted(ts)[len(ts)//2]), "checksum", int(p.to(torch.int64).sum()))
