import numpy as np, sys
sys.path.insert(0,'.')
from clustering_amd.synth import gaussian_blobs
n,d=int(sys.argv[1]),int(sys.argv[2]); r=float(sys.argv[3])
c=gaussian_blobs(n,d)
def tiles_boxes(order):
    cs=c[order]; T=n//32
    cs=cs[:T*32].reshape(T,32,d)
    return cs.min(1), cs.max(1)
def frac(lo,hi,nq=300,dims=None):
    T=lo.shape[0]; rng=np.random.default_rng(1); q=rng.choice(T,nq,replace=False)
    if dims is None: dims=range(d)
    dims=list(dims)
    tot=0
    for t in q:
        g=np.maximum(0,np.maximum(lo[t,dims]-hi[:,dims], lo[:,dims]-hi[t,dims]))
        tot+=((g*g).sum(1) < r*r).sum()
    return tot/(nq*T)
# current: 2-D cell key (cell = r) row-major
mn=c.min(0)
key=np.floor((c[:,0]-mn[0])/r).astype(np.int64)*100000+np.floor((c[:,1]-mn[1])/r).astype(np.int64)
o=np.argsort(key,kind='stable'); lo,hi=tiles_boxes(o)
print("2-D cells, 2-D boxes   :", frac(lo,hi,dims=[0,1]))
print("2-D cells, D-dim boxes :", frac(lo,hi))
# Morton over all dims, quantile bins
def morton(bits, dims):
    codes=np.zeros(n,dtype=np.uint64)
    q=[]
    for k in dims:
        # quantile binning
        ranks=np.argsort(np.argsort(c[:,k],kind='stable'),kind='stable')
        q.append((ranks*(1<<bits)//n).astype(np.uint64))
    for b in range(bits-1,-1,-1):
        for j,k in enumerate(dims):
            codes=(codes<<np.uint64(1))|((q[j]>>np.uint64(b))&np.uint64(1))
    return codes
for bits in (2,3):
    codes=morton(bits, range(d))
    o=np.argsort(codes,kind='stable'); lo,hi=tiles_boxes(o)
    print(f"Morton {bits} bits x {d} dims (quantile), D-dim boxes:", frac(lo,hi))
# cluster-aware: first 2 dims coarse (cell key) then Morton on the rest
codes=morton(3, range(d))
for dims,bits in (((0,1,2),5),((0,1,2,3),4),((0,1,2,3,4),3),((2,3,4),5)):
    codes=morton(bits, dims)
    o=np.argsort(codes,kind='stable'); lo,hi=tiles_boxes(o)
    print(f"Morton {bits} bits x dims {dims}, D-dim boxes:", frac(lo,hi))
