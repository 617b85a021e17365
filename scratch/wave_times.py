"""per-wave durations of nn_pruned_kernel in one segment of eight (a build with -DDC_WAVE_STAMPS, DC_LIB_PATH; the stamps are no longer in the product sources: `git apply -p0 scratch/r6_wave_stamps.patch` first):
how much of the launch is its end (slots idle behind the last long waves) and how uneven the waves are"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens, capi
from clustering_amd.synth import gaussian_blobs
n, d, G = 1000000, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 8
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
pops = dens.calculate_populations_partial(c, [0.2])
fe = dens.calculate_free_energies(pops[0].contiguous())
dens.sweep_timing(True)
for seg in ((0, 3) if G > 1 else (0,)):
    dens.calculate_populations_segment(c, [0.2], seg, G)
    dens.nearest_neighbors_segment(c, fe, seg, G, stats_valid=True) if G > 1 else dens.nearest_neighbors_partial(c, fe, stats_valid=True)
    torch.cuda.synchronize()
    kms = dens.last_sweep_ms("nn", c.device)
    N = 1 << 17
    buf = np.zeros((N, 10), dtype=np.uint64)
    rc = capi.lib.dc_dbg_wave_times(buf.ctypes.data_as(C.c_void_p), C.c_size_t(N))
    assert rc == 0
    live = buf[:, 1] > 0
    t0, t1, ch = buf[live, 0].astype(np.int64), buf[live, 1].astype(np.int64), (buf[live, 2] >> np.uint64(32)).astype(np.int64)
    dur = (t1 - t0) / 100.0   # us (100 MHz)
    start, end = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0
    span = end.max()
    slots = 2048
    np.savez_compressed(f"gpurun_out/nn_waves_G{G}_seg{seg}.npz", idx=np.flatnonzero(live), dur=dur, ch=ch, grp=(buf[live, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), start=start)
    print(f"segment {seg}: kernel {kms*1e3:.0f} us, waves {live.sum()}, span {span:.0f} us, sum of wave times / {slots} slots = {dur.sum()/slots:.0f} us "
          f"(occupancy {dur.sum()/slots/span:.2f}); wave us: mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; "
          f"chains/wave mean {ch.mean():.0f} max {ch.max()}; us per chain {dur.sum()/max(ch.sum(),1):.3f}")
    t_set = (buf[live, 3].astype(np.int64) - t0) / 100.0
    scan = (buf[live, 4] >> np.uint64(32)).astype(np.int64) / 100.0
    fl = (buf[live, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64) / 100.0
    rings = (buf[live, 5] >> np.uint64(32)).astype(np.int64)
    on = (buf[live, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    print(f"   per wave (mean us): set-up {t_set.mean():.1f}, box scans {scan.mean():.1f}, candidate flushes at ring ends {fl.mean():.1f}, "
          f"rest (chains + candidate path) {(dur - t_set - scan - fl).mean():.1f}; rings {rings.mean():.2f}, chains that went on {on.sum()/max(ch.sum(),1):.3f}")
    # the path behind the early-out test (shader cycles, clock64): chains computed again in full, their epilogues, the
    # parking of candidates and the flushes inside it
    cyc_tot = (buf[live, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cyc_on = (buf[live, 6] >> np.uint64(32)).astype(np.int64)
    cyc_fl = (buf[live, 7] >> np.uint64(32)).astype(np.int64)
    nfl = (buf[live, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    iters = (buf[live, 8] >> np.uint64(32)).astype(np.int64)
    cands = (buf[live, 8] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    mhz = cyc_tot.sum() / dur.sum()
    print(f"   clock64: {mhz:.0f} cycles per us; behind the early-out test {cyc_on.sum()/cyc_tot.sum():.3f} of the wave time "
          f"({cyc_on.sum()/max(on.sum(),1):.0f} cycles per chain that went on), of which flushes {cyc_fl.sum()/cyc_tot.sum():.3f} "
          f"({nfl.sum()/max(on.sum(),1):.3f} flushes, {cands.sum()/max(on.sum(),1):.2f} candidates, {iters.sum()/max(on.sum(),1):.2f} chains into the per-element path per such chain; "
          f"{cyc_fl.sum()/max(nfl.sum(),1):.0f} cycles per flush)")
    sA = ((buf[live, 9] >> np.uint64(40)) & np.uint64(0xFFFFF)).astype(np.int64) / 100.0
    sB = ((buf[live, 9] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.int64) / 100.0
    sC = (buf[live, 9] & np.uint64(0xFFFFF)).astype(np.int64) / 100.0
    n_groups_dbg = max(1, len(np.unique((buf[live, 2] & np.uint64(0xFFFFFFFF)))))
    idx = np.flatnonzero(live)
    share0 = idx < n_groups_dbg
    print(f"   set-up of a wave (mean us since its start): query operands / norms / free energies / published words loaded {sA.mean():.1f}, "
          f"query rows staged {sB.mean():.1f}, seeds done {sC.mean():.1f} (first share: {sC[share0].mean():.1f}, others: {sC[~share0].mean():.1f}), first ring's scan starts {t_set.mean():.1f}")
    # by launch order (the reference share is the slow grid dimension: the first waves run without any published bound)
    idx = np.flatnonzero(live)
    rest = dur - t_set - scan - fl
    for lo_, hi_ in ((0, 0.1), (0.1, 0.3), (0.3, 0.6), (0.6, 1.0)):
        m = (idx >= lo_ * idx.max()) & (idx < hi_ * idx.max() + 1)
        print(f"   waves {lo_:.0%} - {hi_:.0%} of the launch order: us per chain {rest[m].sum()/max(ch[m].sum(),1):.4f}, went on {on[m].sum()/max(ch[m].sum(),1):.3f}, set-up {t_set[m].mean():.1f} us, chains/wave {ch[m].mean():.0f}")
    # how many waves are still running in the last 10 / 20 / 30 % of the span
    for f in (0.7, 0.8, 0.9):
        print(f"   running at {f:.0%} of the span: {((start < f*span) & (end > f*span)).sum()} waves")
    order = np.argsort(-dur)[:5]
    print("   longest waves: ", [(round(float(dur[i]),1), int(ch[i]), round(float(start[i]),0)) for i in order])
    wv = (buf[live, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    for i in np.argsort(-dur)[:8]:
        print(f"      wave id {idx[i]} (group {wv[i]}, share {idx[i] // max(1, (idx.max() + 1) // max(1, int(round((idx.max()+1) / max(1, len(np.unique(wv)))))))}): {dur[i]:.0f} us, chains {ch[i]}, went on {on[i]}, "
              f"rings {rings[i]}, candidates {cands[i]}, flushes {nfl[i]}, per-element path {iters[i]}, behind the test {cyc_on[i]/max(cyc_tot[i],1):.2f} of its time, flush {cyc_fl[i]/max(cyc_tot[i],1):.2f}")
    # the groups by their total time over all shares
    tot = {}
    for g_, d_ in zip(wv, dur): tot[g_] = tot.get(g_, 0.0) + d_
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:8]
    print("   groups by wave time over all their shares (us):", [(int(g_), round(t_)) for g_, t_ in top], " mean", round(float(np.mean(list(tot.values())))))
