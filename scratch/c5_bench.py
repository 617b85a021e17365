"""C5 shape (BASELINE.json configs[4]: 5M x 30, 8 radii, 8 GPUs): what ONE rank of the 8-GPU run computes
(segment 3 of 8 of the sweep's spatial order), timed per phase with HIP events on the launch stream, as one
bench-style JSON line.  Used under rocprofv3 by scratch/profile_r2.sh (kernel stats + TCC_EA0 counters)."""
import argparse, json, sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import density as dens
from clustering_amd.synth import gaussian_blobs
ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=5_000_000)
ap.add_argument('--d', type=int, default=30)
ap.add_argument('--segments', type=int, default=8)
ap.add_argument('--segment', type=int, default=3)
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--pop-only', action='store_true')
ap.add_argument('--radii', type=float, nargs='+', default=[0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 0.65])
a = ap.parse_args()
n, d, G = a.n, a.d, a.segments
c = torch.from_numpy(gaussian_blobs(n, d)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def timed(fn):
    ts = []
    for _ in range(a.reps):
        ev[0].record(); out = fn(); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    return out, min(ts)
p, pop_ms = timed(lambda: dens.calculate_populations_segment(c, a.radii, a.segment, G))
pop_tiles = dens.evaluated_tiles(c.device)[0]
pop_mfma = dens.issued_mfmas(c.device)[0]
comp_info = dens.components_info(c)
if a.pop_only:
    pl = torch.stack([x.to(torch.int64) for x in p])
    w = torch.arange(1, pl.shape[1] + 1, device=pl.device, dtype=torch.int64) % 1000003
    print(json.dumps({'pop_8_radii_ms': pop_ms, 'tile_pairs': pop_tiles, 'sum': int(pl.sum().item()), 'wsum': int((pl * w).sum().item())}))
    sys.exit(0)
# FE needs the populations of all rows: one full single-radius sweep here (a real run all-reduces the segments)
pf, full_ms = timed(lambda: dens.calculate_populations_partial(c, [a.radii[len(a.radii) // 2]]))
full_tiles = dens.evaluated_tiles(c.device)[0]
fe = dens.calculate_free_energies(pf[0].contiguous())
nn, nn_ms = timed(lambda: dens.nearest_neighbors_segment(c, fe, a.segment, G))
nn_tiles = dens.evaluated_tiles(c.device)[1]
nn_mfma = dens.issued_mfmas(c.device)[1]
nm = (3 * d + 2 + 15) // 16
rows = int((p[0] != 0).sum().item())
def roof(tiles, mfmas, ms):
    # executed = the v_mfma_f32_32x32x16_f16 instructions the kernel ISSUED (its own counter) x 32768 flop: the neighbour
    # sweep's early-out leaves most chains at their coarse MFMAs (round 4 charged NM per tile pair: 0.925 against a
    # measured matrix-pipe busy of 0.33)
    return {"tile_pairs": tiles, "mfma_issued": mfmas, "mfma_per_tile_pair": mfmas / max(tiles, 1),
            "algorithmic_tflops": tiles * 1024.0 * 2 * d / (ms * 1e-3) / 1e12,
            "frac_algorithmic_f16_peak_2500": tiles * 1024.0 * 2 * d / (ms * 1e-3) / 2.5e15,
            "executed_tflops": mfmas * 32768.0 / (ms * 1e-3) / 1e12,
            "frac_executed_f16_peak_2500": mfmas * 32768.0 / (ms * 1e-3) / 2.5e15}
line = {
    "workload": f"{n} x {d}, radii {a.radii}, segment {a.segment} of {G} (one rank of the 8-GPU run)",
    "rows_of_the_segment": rows, "mfma_per_tile_pair": nm, "components": comp_info,
    "pop_8_radii_ms": pop_ms, "pop_per_radius_ms": pop_ms / len(a.radii), "nn_ms": nn_ms,
    "full_single_radius_sweep_all_rows_ms": full_ms,
    "frame_pairs_per_s_this_rank": {"pop": len(a.radii) * float(rows) * n / (pop_ms * 1e-3), "nn": float(rows) * n / (nn_ms * 1e-3)},
    "evaluated_fraction": {"pop (mean over radii)": pop_tiles * 1024.0 / (len(a.radii) * float(rows) * n),
                           "nn": nn_tiles * 1024.0 / (float(rows) * n),
                           "full sweep": full_tiles * 1024.0 / (float(n) * n)},
    "roofline_pop": roof(pop_tiles, pop_mfma, pop_ms), "roofline_nn": roof(nn_tiles, nn_mfma, nn_ms),
    "note_pop": "eight radii in ONE symmetric sweep (pop_msym_kernel<6, 8, in place>): every unordered tile pair once for all radii, both "
                "frames credited, the thresholds of radii 1..7 taken off the accumulator by rank-1 MFMAs (mfma_issued counts them); the time "
                "is the per-radius epilogue and the reference-side bookkeeping (~250 vector instructions per tile pair), not the matrix pipe",
    "hbm_model": {"Q_res": rows, "note": "all query rows of the rank are resident in one launch (TQ*32 per wave, every wave "
                  "streams the surviving reference tiles), so the streamed model of SURVEY 8(d) is one pass over the coordinates",
                  "algorithmic_bytes_per_sweep": n * d * 4 + rows * 16,
                  "operand_image_bytes": ((n + 31) // 32) * nm * 1024},
}
print(json.dumps(line))
