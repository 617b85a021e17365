"""Time of one segment of a G-way sharded run (what one rank of bench.py --gpus G computes), per phase."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
n, d, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tp, tn, tf = [], [], []
for seg in list(range(G)) * 2:
    ev[0].record(); dens.calculate_populations_segment(c, [0.2], seg, G)
    ev[1].record(); dens.calculate_free_energies(pops[0].contiguous())
    ev[2].record(); dens.nearest_neighbors_segment(c, fe, seg, G)
    ev[3].record(); torch.cuda.synchronize()
    tp.append(ev[0].elapsed_time(ev[1])); tf.append(ev[1].elapsed_time(ev[2])); tn.append(ev[2].elapsed_time(ev[3]))
tp, tf, tn = np.array(tp[G:]), np.array(tf[G:]), np.array(tn[G:])
print("   nn per segment:", np.round(tn, 2), " pop:", np.round(tp, 2))
print(f"G={G}: pop seg mean {tp.mean():.2f} max {tp.max():.2f} ms | fe {tf.mean():.2f} | nn seg mean {tn.mean():.2f} max {tn.max():.2f} ms | sum of maxima {tp.max()+tf.mean()+tn.max():.2f} ms")
