"""What another launch order or uneven reference shares would buy: list scheduling of the MEASURED per-wave durations
(gpurun_out/*_waves_G*_seg*.npz, written by scratch/wave_times.py / pop_wave_times.py on a -DDC_WAVE_STAMPS build) on the
chip's 2 048 wave slots."""
import glob, heapq
import numpy as np


def sched(durs, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for d in durs:
        t = heapq.heappop(h) + d
        end = max(end, t)
        heapq.heappush(h, t)
    return end


for f in sorted(glob.glob('gpurun_out/*_waves_G*_seg*.npz')):
    z = np.load(f)
    o = np.argsort(z['idx'])
    dur, grp = z['dur'][o], z['grp'][o]
    ng = len(np.unique(grp))
    share = np.arange(len(dur)) // ng
    ns = share.max() + 1
    gw = {g: dur[grp == g].mean() for g in np.unique(grp)}
    w = np.array([gw[g] for g in grp])
    setup = 13.0 if 'nn_' in f else 8.0
    res = {"as launched (model)": sched(dur), "ideal (sum / slots)": dur.sum() / 2048,
           "heavy groups first inside each share": sched(dur[np.lexsort((-w, share))]),
           "heavy groups first, all shares of a group together": sched(dur[np.lexsort((share, -w))]),
           "longest wave first (oracle)": sched(dur[np.argsort(-dur)])}
    for name, fs in (("shares shrinking linearly 1.5 .. 0.5", np.linspace(1.5, 0.5, ns)),
                     ("last third of the shares half as large", np.r_[np.ones(ns - ns // 3), np.ones(ns // 3) * 0.5])):
        fs = fs / fs.mean()
        res[name] = sched(setup + (dur - setup) * fs[share])
    print(f.split('/')[-1], f"({len(dur)} waves, {ns} shares):", "; ".join(f"{k} {v:.0f}" for k, v in res.items()))
