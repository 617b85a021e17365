import sys, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
from oracle.oracle import Oracle
o = Oracle()
rng = np.random.default_rng(5)
base = gaussian_blobs(700, 10, seed=9)
c = np.concatenate([base, base[:300], base[100:150]]).astype(np.float32)
c = c[rng.permutation(c.shape[0])]
ct = torch.from_numpy(c).cuda()
want = o.populations(c, [0.2, 0.25])
fe = o.free_energies(want[0])
exp = o.nearest_neighbors(c, fe)
fet = torch.from_numpy(fe).cuda()
for variant in ("direct", "mfma"):
    nn = [t.cpu().numpy() for t in dens.nearest_neighbors_partial(ct, fet, variant=variant)]
    bad = np.nonzero(nn[0].astype(np.uint64) != exp[0])[0]
    print(variant, "nn mismatches", len(bad))
    for i in bad[:10]:
        print(" row", i, "got", nn[0][i], nn[1][i], "exp", exp[0][i], exp[1][i],
              "d2(got)", o.dist2(c[i], c[nn[0][i]]) if nn[0][i] < len(c) else None, "tile", i // 32, "gottile", nn[0][i] // 32, "exptile", exp[0][i] // 32)
    bad = np.nonzero(nn[2].astype(np.uint64) != exp[2])[0]
    print(variant, "hd mismatches", len(bad))
    for i in bad[:10]:
        print(" row", i, "got", nn[2][i], nn[3][i], "exp", exp[2][i], exp[3][i])
