"""per-wave durations of pop_pruned_kernel (a -DDC_WAVE_STAMPS build, DC_LIB_PATH; the stamps are no longer in the product sources: `git apply -p0 scratch/r6_wave_stamps.patch` first): occupancy of the wave slots and the
longest waves of a launch, for all rows (G = 1) or one segment of eight"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens, capi
from clustering_amd.synth import gaussian_blobs
n, d, G = 1000000, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 8
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
dens.calculate_populations_partial(c, [0.2])
dens.sweep_timing(True)
for seg in ((0, 3) if G > 1 else (0,)):
    dens.calculate_populations_segment(c, [0.2], seg, G) if G > 1 else dens.calculate_populations_partial(c, [0.2])
    torch.cuda.synchronize()
    kms = dens.last_sweep_ms("pop", c.device)
    N = 1 << 17
    buf = np.zeros((N, 3), dtype=np.uint64)
    assert capi.lib.dc_dbg_pop_wave_times(buf.ctypes.data_as(C.c_void_p), C.c_size_t(N)) == 0
    live = buf[:, 1] > 0
    # (the stamps of earlier launches stay in the table: only the waves of the last one)
    live &= buf[:, 0].astype(np.int64) >= buf[:, 1].astype(np.int64).max() - int(kms * 1.3e5)
    idx = np.flatnonzero(live)
    t0, t1 = buf[live, 0].astype(np.int64), buf[live, 1].astype(np.int64)
    ch = (buf[live, 2] >> np.uint64(32)).astype(np.int64); grp = (buf[live, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    dur = (t1 - t0) / 100.0
    start, end = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0
    span = end.max()
    print(f"segment {seg}: kernel {kms*1e3:.0f} us, waves {live.sum()} ({len(np.unique(grp))} groups), span {span:.0f} us, sum of wave times / 2048 slots = {dur.sum()/2048:.0f} us "
          f"(occupancy {dur.sum()/2048/span:.2f}); wave us: mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; "
          f"chains/wave mean {ch.mean():.0f} max {ch.max()}; us per chain {dur.sum()/max(ch.sum(),1):.3f}")
    np.savez_compressed(f"gpurun_out/pop_waves_G{G}_seg{seg}.npz", idx=idx, dur=dur, ch=ch, grp=grp, start=start)
    for f in (0.5, 0.7, 0.8, 0.9, 0.95):
        print(f"   running at {f:.0%} of the span: {((start < f*span) & (end > f*span)).sum()} waves")
    for i in np.argsort(-dur)[:6]:
        print(f"      wave id {idx[i]} (group {grp[i]}): {dur[i]:.0f} us, chains {ch[i]}, started at {start[i]:.0f} us, {dur[i]/max(ch[i],1):.3f} us per chain")
    # by launch order: us per chain
    for lo_, hi_ in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
        m = (idx >= lo_ * idx.max()) & (idx < hi_ * idx.max() + 1)
        print(f"   waves {lo_:.0%} - {hi_:.0%} of the launch order: us per chain {dur[m].sum()/max(ch[m].sum(),1):.4f}, chains/wave {ch[m].mean():.0f}, mean start {start[m].mean():.0f} us")
