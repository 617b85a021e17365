#!/usr/bin/env python3
"""bench.py -- the `clustering density` hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one synthetic trajectory:
    populations (one sweep over all radii) -> free energies -> nearest neighbour / nearest
    neighbour with lower free energy, with coordinates already resident in HBM and all outputs
    left resident in HBM (after the RCCL all-reduce / all-gather when N > 1).
Workload (BASELINE.json configs[2], the one the metric is quoted on; it fits one GPU):
    1 000 000 frames x 10 dims, 3-Gaussian-blob generator of SURVEY.md 8(d) (seed 20240), r = 0.2.
Metric: frame-pairs/s (density pop+nn) = 2*N^2 / t_step  (ordered pairs of both sweeps per second).
Prints ONE JSON line on rank 0 (contract in the task statement), including
    "roofline":     the dominant sweep kernel against the pipe it runs on (the f16 MFMA pipe, dense peak
                    2.5 PFLOP/s).  "frac" is SURVEY.md 8(d)'s ALGORITHMIC figure: 2*D flop per EVALUATED frame
                    pair / the kernel's own duration (HIP events around the kernel launch, on its stream) / peak.
                    "frac_executed" counts what the pipe executes: the v_mfma_f32_32x32x16_f16 instructions the kernel
                    ISSUED (its own counter, dc_hip_workspace_mfma_counters_dev; = SQ_VALU_MFMA_BUSY_CYCLES / 32 of a
                    counter profile) x 32*32*16*2 flop -- zero-padded K slots and the three piece products included, the
                    MFMAs the neighbour sweep's early-out leaves out NOT included; "fp32_equivalent" prices the
                    algorithmic flops against the fp32 MFMA peak 157.3 TFLOP/s (BASELINE.json's target figure; it
                    exceeds 1 because the contraction does not run on that pipe).  Pruned-away pairs earn no
                    credit anywhere.  "roofline_by_kernel" carries the same for BOTH sweeps.
    "phases_ms":    the step split into pop_prep / pop_kernel / pops_allreduce / fe / nn_prep / nn_kernel /
                    nn_merge, max and min over the ranks (measured in separate instrumented steps).
    "fp32_mfma_instance": BASELINE.json's literal target -- the fp32-input MFMA variant (v_mfma_f32_32x32x2_f32, every
                    pair) of the same workload, one warm-up + two steps outside the timed region: fraction of the fp32
                    MFMA roof per sweep (target >= 0.60).
    "cpu_baseline": the CPU restatement (oracle, fast build, all host threads) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md, matrix cores: BF16/F16 ~2.5 PF dense


def n_mfma(d):
    """v_mfma_f32_32x32x16_f16 per 32x32 tile pair: 3 piece products per column + 2 constant slots on
    the K axis, 16 slots per MFMA (dc_mfma_kernels.hpp: nm_for)."""
    return (3 * d + 2 + 15) // 16


def executed_flop_per_pair(d):
    """what the f16 matrix pipe executes per frame pair, zero-padded K slots included: NM * 16 * 2"""
    return 32 * n_mfma(d)


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks ourselves (one process per GPU, RCCL), as a
    CHILD process and before this process has touched torch or HIP, relay its output and exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


PMC_PROFILES = ("r6_c3_pmc.json",)   # counter summaries of THIS tree's kernels (older ones describe other kernels)


def csrc_digest():
    """Digest of the product's sources as they lie in the tree (clustering_amd/csrc/digest.py: csrc + include/,
    comments left out -- a comment edit does not orphan the profiles).  The library embeds the digest of the sources it
    was BUILT from (library_digest()); the two differ when the tree was edited after the build."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dc_digest", os.path.join(ROOT, "clustering_amd", "csrc", "digest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_digest()


def library_digest():
    """what the loaded libdcdensity.so says it was built from (dc_hip_build_digest): the binary the timed run executes"""
    from clustering_amd import capi
    return capi.lib.dc_hip_build_digest().decode()


def measured_counters(kernel, n, d, radii, variant):
    """Counter figures of a sweep kernel (memory-side bytes per launch, matrix-pipe busy, VALU
    instructions per tile pair): not measurable from inside the timed run (PMC counters need their own
    rocprofv3 passes), so they come from the committed counter summary of the SAME workload
    (profiles/r*_pmc.json, produced with scratch/pmc_summary.py / make_pmc_profile.py, which records the
    commit of the kernels it measured); {} for any other workload or variant."""
    if variant not in ("auto", "pruned"):
        return {}
    here = library_digest()   # counters describe a BINARY: the one this run loaded
    for fname in PMC_PROFILES:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", fname)))
        except (OSError, ValueError):
            continue
        w = prof.get("workload", {})
        if (w.get("n_rows"), w.get("n_cols"), w.get("radii")) != (n, d, list(radii)):
            continue
        if prof.get("csrc_digest") != here:
            # REFUSED: the counters were collected on other kernel sources than the ones this run times
            return {"source": "profiles/" + fname, "source_commit": prof.get("commit"), "stale": True,
                    "source_digest": prof.get("csrc_digest"), "digest_here": here}
        tag = "pop_pruned_kernel" if kernel == "population_count" else "nn_pruned_kernel"
        for name, e in prof.get("kernels", {}).items():
            if tag in name:
                return {"traffic": e.get("traffic_bytes"), "mfma_busy": e.get("matrix_pipe_utilisation"),
                        "valu_insts_per_tile_pair": e.get("valu_insts_per_32x32_tile_pair"),
                        "source": "profiles/" + fname, "source_commit": prof.get("commit"), "stale": False,
                        "source_digest": here, "digest_here": here}
    return {}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-rows", type=int, default=1_000_000)
    ap.add_argument("--n-cols", type=int, default=10)
    ap.add_argument("--radii", type=float, nargs="+", default=[0.2])
    ap.add_argument("--variant", default="auto", choices=["auto", "direct", "mfma", "pruned", "mfma32"])
    ap.add_argument("--cpu-sample", type=int, default=150000,
                    help="rows of the workload the CPU baseline is timed on (0 = skip)")
    ap.add_argument("--no-nn", action="store_true", help="populations + free energies only (C2-style)")
    ap.add_argument("--no-full-sweep", action="store_true", help="skip the unpruned reference sweeps (roofline_full_sweep)")
    ap.add_argument("--no-fp32-instance", action="store_true", help="skip the fp32-input MFMA instance (fp32_mfma_instance)")
    return ap.parse_args()


def cpu_baseline(coords_np, radii, sample_rows, want_nn):
    """Times the CPU restatement (oracle/dc_oracle.c, DCO_FAST build: -O3 -ffast-math -mavx2 -mfma,
    OpenMP over all host threads; box-grid pruned i<j populations + brute-force neighbours -- the
    reference's own algorithm, density_clustering.cpp:126-288) on the first sample_rows rows."""
    from oracle.oracle import Oracle, build
    build()
    o = Oracle(fast=True)
    c = coords_np[:sample_rows]
    n = c.shape[0]
    t0 = time.perf_counter()
    pops = o.populations(c, radii, boxgrid=True)
    t1 = time.perf_counter()
    fe = o.free_energies(pops[0])
    t2 = time.perf_counter()
    if want_nn:
        o.nearest_neighbors(c, fe)
    t3 = time.perf_counter()
    sweeps = 2 if want_nn else 1
    return {
        "value": sweeps * float(n) * n / (t3 - t0),
        "unit": "frame-pairs/s",
        "cores": o.threads,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"first {n} rows of the workload ({n}x{c.shape[1]}, radii {list(radii)}): "
                  f"pops {t1 - t0:.2f}s (box grid, i<j) + fe {t2 - t1:.3f}s"
                  + (f" + nn {t3 - t2:.2f}s (brute force)" if want_nn else "")
                  + "; rate = sweeps*n^2/t at THIS n (the pruned pop sweep gets relatively cheaper as n grows)",
    }


PHASES = ("pop_prep", "pop_kernel", "pops_allreduce", "fe", "nn_prep", "nn_kernel", "nn_merge")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the density path has no CPU fallback")
    # DC_BENCH_ONE_DEVICE=1 (development only): all ranks share cuda:0 and talk over gloo, so that the
    # sharded path can be exercised end to end on a single-GPU box; the timings mean nothing then.
    one_device = os.environ.get("DC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from clustering_amd import density
    from clustering_amd.distributed import HipBackend, ShardedDensity
    from clustering_amd.rows import shard_rows
    from clustering_amd.synth import gaussian_blobs

    n, d = args.n_rows, args.n_cols
    want_nn = not args.no_nn
    coords_np = gaussian_blobs(n, d)            # same seed on every rank: coordinates are replicated
    coords = torch.from_numpy(coords_np).to(dev)
    backend = HipBackend(args.variant)
    # (the layout verdict of an unpack is that of the LAST unpack only: asked for after the last timed step and after every
    #  instrumented step below -- inside the timed loop it would add a host synchronisation per step)
    job = ShardedDensity(backend, check_layout=False)
    lo, hi = shard_rows(n, world, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.run(coords, args.radii, 0, want_nn)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = job.run(coords, args.radii, 0, want_nn)
    barrier()
    t1 = time.perf_counter()
    if world > 1:
        job.check_layouts(dev)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # ---- instrumented steps (outside the whole-job timed region): the SAME ShardedDensity.run with phase marks
    # (torch events on the current stream, which is the stream the C ABI launches on) and the library's own event
    # pair around the main sweep kernels (dc_hip_sweep_timing) -- prep = call - kernel
    density.sweep_timing(True)
    reps = max(1, min(args.steps, 3))
    acc = {k: [] for k in PHASES}
    for _ in range(reps):
        marks = []

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        job.run(coords, args.radii, 0, want_nn, mark=mark)
        torch.cuda.synchronize()
        if world > 1:
            job.check_layouts(dev)
        span = {}
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            span[name] = a.elapsed_time(b)
        pop_k = density.last_sweep_ms("pop", dev) if args.variant != "direct" and d <= 64 else span.get("pop", 0.0)
        nn_k = 0.0
        if want_nn:
            nn_k = density.last_sweep_ms("nn", dev) if args.variant != "direct" and d <= 64 else span.get("nn", 0.0)
        acc["pop_kernel"].append(pop_k)
        acc["pop_prep"].append(max(0.0, span.get("pop", 0.0) - pop_k))
        acc["pops_allreduce"].append(span.get("pops_allreduce", 0.0))
        acc["fe"].append(span.get("fe", 0.0))
        acc["nn_kernel"].append(nn_k)
        acc["nn_prep"].append(max(0.0, span.get("nn", 0.0) - nn_k))
        acc["nn_merge"].append(span.get("nn_merge", 0.0))
    density.sweep_timing(False)
    mine = torch.tensor([float(np.mean(acc[k])) for k in PHASES], dtype=torch.float64, device=dev)
    ph_max, ph_min = mine.clone(), mine.clone()
    if world > 1:
        dist.all_reduce(ph_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(ph_min, op=dist.ReduceOp.MIN)
    pop_t, nn_t = float(mine[PHASES.index("pop_kernel")]) * 1e-3, float(mine[PHASES.index("nn_kernel")]) * 1e-3

    # pairs actually evaluated (the pruned variants skip tile pairs that are provably too far apart):
    # each sweep leaves its 32x32-tile count in the workspace header
    pops_c = (backend.populations_segment(coords, args.radii, rank, world) if world > 1
              else backend.populations_partial(coords, args.radii, lo, hi))
    pop_tiles = density.evaluated_tiles(dev)[0]
    pop_mfma = density.issued_mfmas(dev)[0]
    nn_tiles = nn_mfma = 0
    if want_nn:
        if world > 1:
            dist.all_reduce(pops_c, op=dist.ReduceOp.SUM)
        fe_c = backend.free_energies(pops_c[0].contiguous())
        if world > 1:
            backend.nearest_neighbors_segment(coords, fe_c, rank, world)
        else:
            backend.nearest_neighbors_partial(coords, fe_c, lo, hi)
        nn_tiles = density.evaluated_tiles(dev)[1]
        nn_mfma = density.issued_mfmas(dev)[1]
    # reference point for the roofline: the same sweeps with EVERY pair evaluated (DC_VARIANT_MFMA)
    full_ms = None
    if args.variant in ("auto", "pruned") and rank == 0 and not args.no_full_sweep:
        fb = HipBackend("mfma")
        e0, e1, e2, e3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        fb.populations_partial(coords, args.radii, lo, hi)          # warm-up (workspace, code)
        e0.record()
        pf = fb.populations_partial(coords, args.radii, lo, hi)
        e1.record()
        full_ms = {"pop_kernel": None, "nn_kernel": None}
        if want_nn:
            ff = fb.free_energies(pf[0].contiguous())
            e2.record()
            fb.nearest_neighbors_partial(coords, ff, lo, hi)
            e3.record()
        torch.cuda.synchronize()
        full_ms["pop_kernel"] = e0.elapsed_time(e1)
        if want_nn:
            full_ms["nn_kernel"] = e2.elapsed_time(e3)

    # BASELINE.json's literal target ("an MFMA-tiled fp32 variant ... >= 60 % of fp32 MFMA roofline on 1 x MI355X for
    # N = 1e6 x D = 10"): the fp32-input MFMA instance of the SAME workload (dc_mfma32.hpp: v_mfma_f32_32x32x2_f32, every
    # pair evaluated, one radius per sweep), one warm-up + two steps outside the timed region
    fp32_inst = None
    if (args.variant in ("auto", "pruned") and world == 1 and d in (9, 10) and len(args.radii) == 1
            and not args.no_fp32_instance):
        fb32 = HipBackend("mfma32")
        density.sweep_timing(True)
        tp, tn = [], []
        for it in range(3):
            p32 = fb32.populations_partial(coords, args.radii, 0, n)
            t_pop32 = density.last_sweep_ms("pop", dev)
            t_nn32 = None
            if want_nn:
                f32_ = fb32.free_energies(p32[0].contiguous())
                nn32 = fb32.nearest_neighbors_partial(coords, f32_, 0, n)
                t_nn32 = density.last_sweep_ms("nn", dev)
            if it > 0:
                tp.append(t_pop32)
                if want_nn:
                    tn.append(t_nn32)
        density.sweep_timing(False)
        same = bool((p32 == out["pops"]).all())
        if want_nn:
            same = same and bool((nn32[0] == out["nn_idx"]).all()) and bool((nn32[2] == out["hd_idx"]).all()) \
                and bool((nn32[1].view(torch.int32) == out["nn_d2"].view(torch.int32)).all())
        flop = float(n) * n * 2.0 * d

        def inst(ms_list):
            ms = float(np.mean(ms_list))
            return {"launch_ms": ms, "achieved": flop / (ms * 1e-3) / 1e12, "frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
        fp32_inst = {
            "what": "the fp32-input MFMA instance of this workload (variant mfma32: v_mfma_f32_32x32x2_f32 Gram tiles, EVERY "
                    "ordered pair evaluated, guard band + canonical re-check), 1 warm-up + 2 steps outside the timed region",
            "pipe": "fp32 MFMA (the vector rate on gfx950)", "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "pairs_per_launch": float(n) * n, "flop_per_pair": 2 * d,
            "population_count": inst(tp), "nearest_neighbor_search": inst(tn) if tn else None,
            "frac": min([inst(tp)["frac"]] + ([inst(tn)["frac"]] if tn else [])),
            "frac_definition": "SURVEY.md 8(d): 2*D flop per ordered pair x N^2 / the kernel's launch duration (library event "
                               "pair) / 157.3 TFLOP/s; the smaller of the two sweeps; BASELINE.json's target: >= 0.60",
            "results_equal_default_path": same,
        }
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r6_c3_mfma32_pmc.json")))
            if prof.get("csrc_digest") == library_digest() and (n, d) == (prof.get("n_rows"), prof.get("n_cols")):
                fp32_inst["mfma_busy"] = prof.get("mfma_busy")
                fp32_inst["mfma_busy_source"] = "profiles/r6_c3_mfma32_pmc.json (rocprofv3 --pmc, same library digest)"
        except (OSError, ValueError):
            pass

    sweeps = 2 if want_nn else 1
    pairs_per_step = sweeps * float(n) * float(n)
    ms_per_step = 1e3 * elapsed / args.steps
    value = pairs_per_step / (elapsed / args.steps)
    # tile pairs the sweeps of ALL ranks evaluated in one step (from the kernels' own counters)
    tiles_all = torch.tensor([pop_tiles, nn_tiles], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tiles_all, op=dist.ReduceOp.SUM)
    tiles_all = [int(v) for v in tiles_all.tolist()]

    if rank == 0:
        # Work = ordered (query, reference) pairs the kernel EVALUATED: the 32x32 tile pairs it counted itself (all pairs
        # of the rank's rows for the variants that do not prune); pruned-away pairs earn no roofline credit.
        full_pairs = float(n) * n / world          # this rank's share of the N^2 ordered pairs
        pop_pairs = pop_tiles * 1024.0 if pop_tiles else full_pairs
        nn_pairs = nn_tiles * 1024.0 if nn_tiles else full_pairs
        matrix = args.variant != "direct" and d <= 64

        def roof_of(kernel, pairs, t, mfmas):
            if t <= 0.0:
                return None
            algorithmic = pairs * 2.0 * d / t / 1e12                     # 2*D flop per pair
            # what the f16 pipe does: the MFMA instructions the kernel issued (its own counter) x 32*32*16*2 flop; the
            # sweeps without a counter (every pair: --variant mfma) issue NM per tile pair
            flop_issued = mfmas * 32768.0 if mfmas else pairs * executed_flop_per_pair(d)
            executed = flop_issued / t / 1e12
            pmc = measured_counters(kernel, n, d, args.radii, args.variant) if world == 1 else {}
            if args.variant == "mfma32":
                # the literal fp32-input MFMA instance: K = 2 * ceil(D / 2) slots, one fp32 multiply-add each
                ex32 = pairs * 2.0 * (2 * ((d + 1) // 2)) / t / 1e12
                r = {"bound": "mfma", "pipe": "fp32 (v_mfma_f32_32x32x2_f32: the vector rate on gfx950)", "kernel": kernel,
                     "achieved": algorithmic, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": algorithmic / PEAK_FP32_TFLOPS,
                     "frac_definition": "SURVEY.md 8(d): 2*D fp32 flop per evaluated ordered frame pair / the kernel's launch "
                                        "duration / the fp32 MFMA peak (BASELINE.json's target: >= 0.60)",
                     "flop_per_pair_algorithmic": 2 * d, "achieved_executed": ex32, "frac_executed": ex32 / PEAK_FP32_TFLOPS}
            elif matrix:
                r = {"bound": "mfma", "pipe": "f16 (v_mfma_f32_32x32x16_f16, dense)", "kernel": kernel,
                     "achieved": algorithmic, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": algorithmic / PEAK_BF16_TFLOPS,
                     "frac_definition": "SURVEY.md 8(d): 2*D fp32 flop per EVALUATED ordered frame pair (one multiply-add per "
                                        "dimension) / the kernel's launch duration / the dense peak of the pipe it runs on",
                     "flop_per_pair_algorithmic": 2 * d,
                     "achieved_executed": executed, "frac_executed": executed / PEAK_BF16_TFLOPS,
                     "mfma_issued_per_launch": mfmas or None,
                     "mfma_per_tile_pair": (mfmas / (pairs / 1024.0)) if mfmas else float(n_mfma(d)),
                     "frac_executed_definition": "v_mfma_f32_32x32x16_f16 instructions the kernel issued (counted by the kernel: "
                                                 "dc_hip_workspace_mfma_counters_dev) x 32768 flop / launch duration / peak",
                     "flop_per_pair_executed": flop_issued / pairs,
                     "fp32_equivalent": {"achieved": algorithmic, "peak": PEAK_FP32_TFLOPS,
                                         "frac": algorithmic / PEAK_FP32_TFLOPS,
                                         "what": "the algorithmic flops against the fp32 MFMA/VALU peak (BASELINE.json's "
                                                 "target figure); > 1 is possible because the contraction runs on the f16 pipe"}}
            else:
                r = {"bound": "mfma", "pipe": "fp32 VALU (direct kernels)", "kernel": kernel,
                     "achieved": algorithmic, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": algorithmic / PEAK_FP32_TFLOPS, "flop_per_pair_algorithmic": 2 * d}
            r.update({
                "traffic": pmc.get("traffic"),
                "traffic_note": ("NOT REPORTED: " + str(pmc.get("source")) + " was measured on other kernel sources (digest "
                                 + str(pmc.get("source_digest")) + ", this tree " + str(pmc.get("digest_here"))
                                 + "): regenerate it with scratch/profile_r4.sh. " if pmc.get("stale") else "")
                                + "bytes per launch at the L2's memory side (TCC_EA0 read requests x 128 B + write requests x "
                                "64 B, separate rocprofv3 --pmc pass, Infinity-Cache hits included) from "
                                + str(pmc.get("source")) + " (kernel sources of digest " + str(pmc.get("source_digest"))
                                + ", commit " + str(pmc.get("source_commit"))
                                + "); algorithmic bytes per launch = N*D*4 + outputs = "
                                f"{n * d * 4 + n * 16} B: the kernel is compute-bound and re-streams its operand image through L2",
                "mfma_busy": pmc.get("mfma_busy"),
                "valu_insts_per_tile_pair": pmc.get("valu_insts_per_tile_pair"),
                "pairs_per_launch": pairs,
                "pairs_per_launch_unpruned": full_pairs,
                "launch_ms": 1e3 * t,
                "launch_ms_source": "HIP events recorded by the library around the kernel launch, on its stream "
                                    "(dc_hip_sweep_timing / dc_hip_last_sweep_ms)",
            })
            return r
        by_kernel = {"population_count": roof_of("population_count", pop_pairs, pop_t, pop_mfma)}
        if want_nn:
            by_kernel["nearest_neighbor_search"] = roof_of("nearest_neighbor_search", nn_pairs, nn_t, nn_mfma)
        dom = "nearest_neighbor_search" if (want_nn and nn_t > pop_t) else "population_count"
        roof = dict(by_kernel[dom])
        roof["evaluated_fraction"] = {"pop": pop_pairs / full_pairs, "nn": nn_pairs / full_pairs}
        roof["evaluated_fraction_note"] = ("tile pairs the kernels COMPUTED (their own counters) x 1024 / N^2; the one-radius "
                                           "population sweep computes every unordered pair of query groups once and credits "
                                           "both frames (d2 is symmetric), so its figure covers about twice as many ordered pairs")
        pop_sum = int(out["pops"][0].sum(dtype=torch.int64).item())
        evaluated_all = (tiles_all[0] + tiles_all[1]) * 1024.0 if tiles_all[0] else pairs_per_step
        phases = {k: float(ph_max[i]) for i, k in enumerate(PHASES)}
        phases["sum_of_phases"] = float(sum(phases[k] for k in PHASES))
        phases["unaccounted (host gaps between launches)"] = ms_per_step - phases["sum_of_phases"]
        line = {
            "metric": "frame-pairs/s (density pop+nn)" if want_nn else "frame-pairs/s (density pop)",
            "value": value,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("f32 (Gram form on the fp32-input MFMA, every pair, as a classifier with a guard band; undecided pairs "
                      "re-checked in canonical f32)" if args.variant == "mfma32" else
                      "f32 (Gram form on two fp16 pieces per coordinate on the f16 MFMA pipe, f32 accumulate, as a "
                      "classifier with a guard band; undecided pairs re-checked in canonical f32)"),
            "data": "synthetic",
            "config": {
                "workload": f"{n} frames x {d} dims, 3-Gaussian-blob (sigma 0.08, seed 20240), radii {args.radii}, "
                            + ("pop + free energy + nn/nn_hd" if want_nn else "pop + free energy"),
                "n_rows": n, "n_cols": d, "radii": args.radii, "variant": args.variant,
                "parallelism": f"rows sharded over {world} GPU(s) (every {world}-th query group of the spatial order), coords replicated; "
                               "all-reduce(sum) of the populations + " + job.neighbour_merge_name(),
                "backend": (dist.get_backend() if world > 1 else "none (single process)"),
                "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
            },
            "value_note": "value = sweeps * N^2 / t: ordered pairs ANSWERED per second (pairs pruned away by the "
                          "spatial ordering count, like the reference's box grid skips them); "
                          "evaluated_pairs_per_s = pairs the kernels actually evaluated per second",
            "evaluated_pairs_per_s": evaluated_all / (elapsed / args.steps),
            "phases_ms": phases,
            "phases_ms_min_over_ranks": {k: float(ph_min[i]) for i, k in enumerate(PHASES)},
            "phases_note": "max over the ranks (min beside it) of each phase of instrumented steps run after the timed region: "
                           "*_kernel = the sweep kernel alone (library event pair), *_prep = the rest of the call (statistics, "
                           "orderings, operand images, boxes, unpack), pops_allreduce / nn_merge = the collectives with their "
                           "pack / unpack kernels, fe = free energies (device log + host referee, one stream synchronisation)",
            "check": {"mean_pop_r0": pop_sum / n, "max_pop_r0": int(out["pops"][0].max().item()),
                      "sigma2": (density.compute_sigma2(out["nn_d2"]) if want_nn else None),
                      "reference_run": "BASELINE.md: mean 7233.1, max 65950, sigma2 0.00704766 at C3"},
            "library_digest": library_digest(),
            "source_digest": csrc_digest(),
            "digest_note": "library_digest = dc_hip_build_digest() of the libdcdensity.so this run loaded (sources it was built "
                           "from, comments left out); source_digest = the same digest of the tree; counter figures are taken "
                           "only from profiles whose csrc_digest equals library_digest",
            "roofline": roof,
            "roofline_by_kernel": by_kernel,
            "fp32_mfma_instance": fp32_inst,
        }
        if full_ms is not None:
            # the unpruned sweeps (every ordered pair evaluated) for comparison
            fl = float(n) * n / world * executed_flop_per_pair(d)
            fl32 = float(n) * n / world * 2.0 * d

            def full(ms):
                return {"launch_ms": ms, "frac": fl32 / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "frac_executed": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "fp32_equivalent_frac": fl32 / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
            line["roofline_full_sweep"] = {
                "variant": "mfma (no pruning; launch_ms = whole call)", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "pop": full(full_ms["pop_kernel"]),
                "nn": None if full_ms["nn_kernel"] is None else full(full_ms["nn_kernel"]),
                "frame_pairs_per_s": (sweeps * float(n) * n / world
                                      / (1e-3 * (full_ms["pop_kernel"] + (full_ms["nn_kernel"] or 0.0)))),
            }
        # (rank 0 of a single-GPU run only: the other ranks of a sharded run would wait for it at the barrier)
        if args.cpu_sample > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(coords_np, args.radii, min(args.cpu_sample, n), want_nn)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
