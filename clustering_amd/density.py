"""Host-side Python mirror of the reference's density hot-path interface
(Clustering::Density::CUDA::*, density_clustering_cuda.hpp:13-54) over the C ABI.

torch is used here only as plumbing: device memory (tensors), streams and, in
clustering_amd.distributed, torch.distributed.  All compute goes through
libdcdensity.so; nothing in this package computes distances on the CPU or in torch.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .rows import shard_rows  # noqa: F401  (re-export)

FLT_MAX = float(np.finfo(np.float32).max)


def get_num_gpus():
    """Clustering::Density::CUDA::get_num_gpus (density_clustering_cuda.cu:32-43): raises if none."""
    n = capi.device_count()
    if n == 0:
        raise capi.DensityLibraryError("error: no HIP-compatible GPUs found")
    return n


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    return C.c_void_p(t.data_ptr())


class Workspace:
    """Device scratch for the sweeps (MFMA operand images); grown on demand, reused across calls."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, n_rows, n_cols, n_radii=1):
        need = int(capi.lib.dc_hip_workspace_bytes(n_rows, n_cols, n_radii))
        if need == 0:
            return C.c_void_p(0), 0
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return _dev(self.buf), int(self.buf.numel())


_workspaces = {}


def _workspace(device):
    key = str(device)
    if key not in _workspaces:
        _workspaces[key] = Workspace(device)
    return _workspaces[key]


def _variant(variant, stats_valid):
    """the C ABI's `variant` argument: kernel family | DC_FLAG_STATS_VALID (the workspace header still holds the
    statistics of an earlier sweep over the SAME coordinates: the second call of a populations -> neighbours pair)"""
    return capi.VARIANTS[variant] | (capi.FLAG_STATS_VALID if stats_valid else 0)


def _check_coords(coords):
    if not (isinstance(coords, torch.Tensor) and coords.is_cuda and coords.dtype == torch.float32
            and coords.dim() == 2 and coords.is_contiguous()):
        raise ValueError("coords must be a contiguous float32 CUDA tensor [n_rows, n_cols]")
    return coords.shape[0], coords.shape[1]


def _ascending(rad, out):
    """Several radii go to the library in ASCENDING order (the symmetric multi-radius sweep then leaves out the small
    radii a tile pair holds nothing of -- INTEGRATION.md section 5); the rows come back in the caller's order.
    -> (radii for the call, buffer for the call, row permutation or None)"""
    if rad.size < 2 or bool(np.all(rad[1:] >= rad[:-1])):
        return rad, out, None
    order = np.argsort(rad, kind="stable")
    return np.ascontiguousarray(rad[order]), torch.empty_like(out), torch.from_numpy(order).to(out.device)


def calculate_populations_partial(coords, radii, i_from=0, i_to=None, variant="auto", out=None, stats_valid=False):
    """Per-GPU partial of calculate_populations (density_clustering_cuda.cu:45-137).

    coords: float32 CUDA tensor [n_rows, n_cols]; radii: sequence of float.
    -> torch.int32 [n_radii, n_rows] (the ABI's uint32 bit pattern; populations are <= n_rows
    < 2^31 whenever the tensor itself is addressable), radius-major in the order of ``radii``,
    zero outside [i_from, i_to).
    """
    n_rows, n_cols = _check_coords(coords)
    i_to = n_rows if i_to is None else i_to
    rad = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
    if out is None:
        out = torch.empty((rad.size, n_rows), dtype=torch.int32, device=coords.device)
    assert out.shape == (rad.size, n_rows) and out.dtype == torch.int32 and out.is_contiguous()
    rad_call, dst, order = _ascending(rad, out)
    with torch.cuda.device(coords.device):
        ws, ws_bytes = _workspace(coords.device).get(n_rows, n_cols, rad.size)
        rc = capi.lib.dc_hip_populations_dev(
            _dev(coords), n_rows, n_cols, rad_call.ctypes.data_as(C.POINTER(C.c_float)), rad.size,
            i_from, i_to, _dev(dst), ws, ws_bytes, _variant(variant, stats_valid), _stream_ptr())
    capi.check(rc, "dc_hip_populations_dev")
    if order is not None:
        out[order] = dst
    return out


def calculate_populations_segment(coords, radii, segment, n_segments, variant="auto", out=None, stats_valid=False):
    """Populations of one segment of a sharded run (dc_hip_populations_segment_dev): with the pruned
    sweep every n_segments-th query group of the spatial order, else the reference's row block.  PARTIAL counts
    that merge by summation over the segments (a one-radius pruned sweep is symmetric -- it credits both frames
    of a pair -- so a segment's counts cover all rows; the other sweeps leave zeros outside the segment)."""
    n_rows, n_cols = _check_coords(coords)
    rad = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
    if out is None:
        out = torch.empty((rad.size, n_rows), dtype=torch.int32, device=coords.device)
    assert out.shape == (rad.size, n_rows) and out.dtype == torch.int32 and out.is_contiguous()
    rad_call, dst, order = _ascending(rad, out)
    with torch.cuda.device(coords.device):
        ws, ws_bytes = _workspace(coords.device).get(n_rows, n_cols, rad.size)
        rc = capi.lib.dc_hip_populations_segment_dev(
            _dev(coords), n_rows, n_cols, rad_call.ctypes.data_as(C.POINTER(C.c_float)), rad.size,
            segment, n_segments, _dev(dst), ws, ws_bytes, _variant(variant, stats_valid), _stream_ptr())
    capi.check(rc, "dc_hip_populations_segment_dev")
    if order is not None:
        out[order] = dst
    return out


def nearest_neighbors_segment(coords, fe, segment, n_segments, variant="auto", stats_valid=False):
    """Neighbours of one segment of a sharded run (dc_hip_nearest_neighbors_segment_dev); the rows of
    other segments hold (n_rows+1, FLT_MAX)."""
    n_rows, n_cols = _check_coords(coords)
    assert fe.is_cuda and fe.dtype == torch.float32 and fe.shape == (n_rows,) and fe.is_contiguous()
    dev = coords.device
    nn_idx = torch.empty(n_rows, dtype=torch.int32, device=dev)
    hd_idx = torch.empty(n_rows, dtype=torch.int32, device=dev)
    nn_d2 = torch.empty(n_rows, dtype=torch.float32, device=dev)
    hd_d2 = torch.empty(n_rows, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_bytes = _workspace(dev).get(n_rows, n_cols, 1)
        rc = capi.lib.dc_hip_nearest_neighbors_segment_dev(
            _dev(coords), n_rows, n_cols, _dev(fe), segment, n_segments, _dev(nn_idx), _dev(nn_d2),
            _dev(hd_idx), _dev(hd_d2), ws, ws_bytes, _variant(variant, stats_valid), _stream_ptr())
    capi.check(rc, "dc_hip_nearest_neighbors_segment_dev")
    return nn_idx, nn_d2, hd_idx, hd_d2


def calculate_free_energies(pops):
    """calculate_free_energies (density_clustering.cpp:197-212) for one radius. pops: int32 CUDA [n_rows]."""
    assert pops.is_cuda and pops.dtype == torch.int32 and pops.dim() == 1 and pops.is_contiguous()
    fe = torch.empty(pops.shape[0], dtype=torch.float32, device=pops.device)
    mx = C.c_uint32(0)
    with torch.cuda.device(pops.device):
        rc = capi.lib.dc_hip_free_energies_dev(_dev(pops), pops.shape[0], _dev(fe), C.byref(mx),
                                               _stream_ptr())
    capi.check(rc, "dc_hip_free_energies_dev")
    return fe


def nearest_neighbors_partial(coords, fe, i_from=0, i_to=None, variant="auto", stats_valid=False):
    """Per-GPU partial of nearest_neighbors (density_clustering_cuda.cu:184-284).

    -> (nn_idx int32, nn_d2 float32, hd_idx int32, hd_d2 float32), each [n_rows]; rows outside the
    range hold the reference's "none" value (n_rows+1, FLT_MAX)."""
    n_rows, n_cols = _check_coords(coords)
    i_to = n_rows if i_to is None else i_to
    assert fe.is_cuda and fe.dtype == torch.float32 and fe.shape == (n_rows,) and fe.is_contiguous()
    dev = coords.device
    nn_idx = torch.empty(n_rows, dtype=torch.int32, device=dev)
    hd_idx = torch.empty(n_rows, dtype=torch.int32, device=dev)
    nn_d2 = torch.empty(n_rows, dtype=torch.float32, device=dev)
    hd_d2 = torch.empty(n_rows, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_bytes = _workspace(dev).get(n_rows, n_cols, 1)
        rc = capi.lib.dc_hip_nearest_neighbors_dev(
            _dev(coords), n_rows, n_cols, _dev(fe), i_from, i_to, _dev(nn_idx), _dev(nn_d2),
            _dev(hd_idx), _dev(hd_d2), ws, ws_bytes, _variant(variant, stats_valid), _stream_ptr())
    capi.check(rc, "dc_hip_nearest_neighbors_dev")
    return nn_idx, nn_d2, hd_idx, hd_d2


def evaluated_tiles(device):
    """(pop_tiles, nn_tiles): 32x32 frame-pair tiles evaluated by the last pruned sweeps on this device's
    workspace.  The header is rebuilt by every sweep, so read it right after the sweep of interest
    (the other counter is then 0)."""
    ws = _workspace(device)
    if ws.buf is None:
        return 0, 0
    a, b = C.c_uint64(0), C.c_uint64(0)
    with torch.cuda.device(device):
        rc = capi.lib.dc_hip_workspace_counters_dev(_dev(ws.buf), C.byref(a), C.byref(b), _stream_ptr())
    capi.check(rc, "dc_hip_workspace_counters_dev")
    return int(a.value), int(b.value)


def issued_mfmas(device):
    """(pop, nn): v_mfma_f32_32x32x16_f16 instructions the last pruned sweeps on this device's workspace issued, counted by
    the kernels (dc_hip_workspace_mfma_counters_dev); read right after the sweep of interest, like evaluated_tiles"""
    ws = _workspace(device)
    if ws.buf is None:
        return 0, 0
    a, b = C.c_uint64(0), C.c_uint64(0)
    with torch.cuda.device(device):
        capi.check(capi.lib.dc_hip_workspace_mfma_counters_dev(_dev(ws.buf), C.byref(a), C.byref(b), _stream_ptr()),
                   "dc_hip_workspace_mfma_counters_dev")
    return int(a.value), int(b.value)


def components_info(coords):
    """of the last pruned population sweep over coords on its device: dict(n_components, extent2_global, extent2_local,
    scale) -- dc_hip_workspace_components_dev"""
    n_rows, n_cols = _check_coords(coords)
    ws = _workspace(coords.device)
    n, a, b, s = C.c_uint32(0), C.c_float(0), C.c_float(0), C.c_float(0)
    with torch.cuda.device(coords.device):
        capi.check(capi.lib.dc_hip_workspace_components_dev(_dev(ws.buf), n_rows, n_cols, C.byref(n), C.byref(a),
                                                            C.byref(b), C.byref(s), _stream_ptr()),
                   "dc_hip_workspace_components_dev")
    return {"n_components": int(n.value), "extent2_global": float(a.value), "extent2_local": float(b.value),
            "scale": float(s.value)}


def sweep_timing(enable):
    """dc_hip_sweep_timing: bracket the main sweep kernels with HIP events (measurement aid of bench.py)"""
    capi.check(capi.lib.dc_hip_sweep_timing(1 if enable else 0), "dc_hip_sweep_timing")


def last_sweep_ms(kind, device):
    """duration in ms of the main sweep kernel(s) of one kind ("pop" / "nn") since the last read (synchronises)"""
    ms = C.c_float(0.0)
    with torch.cuda.device(device):
        capi.check(capi.lib.dc_hip_last_sweep_ms(0 if kind == "pop" else 1, C.byref(ms)), "dc_hip_last_sweep_ms")
    return float(ms.value)


def radius_pairs(coords, r2, capacity=None):
    """All unordered frame pairs with canonical d2 < r2 (the radius graph of the reference's screening,
    density_clustering.cpp:292-332) -> (pairs int64 [n_pairs, 2] on the device, pops int32 [n_rows]).
    One counting sweep sizes the buffer unless a capacity is given."""
    n_rows, n_cols = _check_coords(coords)
    dev = coords.device
    pops = torch.empty(n_rows, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)

    def sweep(pairs, cap):
        with torch.cuda.device(dev):
            ws, ws_bytes = _workspace(dev).get(n_rows, n_cols, 1)
            rc = capi.lib.dc_hip_radius_pairs_dev(_dev(coords), n_rows, n_cols, float(r2), _dev(pops),
                                                  _dev(pairs) if pairs is not None else None, cap,
                                                  _dev(count), ws, ws_bytes, _stream_ptr())
        capi.check(rc, "dc_hip_radius_pairs_dev")
        return int(count.item())

    if capacity is None:
        capacity = sweep(None, 0)
    if capacity < 0:
        raise capi.DensityLibraryError("radius pairs need finite coordinates")
    pairs = torch.empty((max(capacity, 1), 2), dtype=torch.int32, device=dev)
    n = sweep(pairs, capacity)
    if n < 0:
        raise capi.DensityLibraryError("radius pairs need finite coordinates")
    if n > capacity:
        return radius_pairs(coords, r2, n)
    return pairs[:n].to(torch.int64), pops


def radius_min_edge(coords, r2, comp, rank, segment=0, n_segments=0):
    """One Boruvka round on the radius graph (dc_hip_radius_min_edge[_segment]_dev): comp, rank int32
    CUDA [n_rows] -> (best int64 [n_rows]: (max rank << 32 | min rank) of the lightest pair leaving
    component id, -1 (all ones) if none; pops int32 [n_rows]).  n_segments > 0: what the queries of one
    segment of a sharded run see (partials merge by unsigned minimum / summation)."""
    n_rows, n_cols = _check_coords(coords)
    dev = coords.device
    for t in (comp, rank):
        assert t.is_cuda and t.dtype == torch.int32 and t.shape == (n_rows,) and t.is_contiguous()
    best = torch.empty(n_rows, dtype=torch.int64, device=dev)
    pops = torch.empty(n_rows, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_bytes = _workspace(dev).get(n_rows, n_cols, 1)
        rc = capi.lib.dc_hip_radius_min_edge_segment_dev(_dev(coords), n_rows, n_cols, float(r2), _dev(comp),
                                                         _dev(rank), segment, n_segments, _dev(best),
                                                         _dev(pops), ws, ws_bytes, _stream_ptr())
    capi.check(rc, "dc_hip_radius_min_edge_segment_dev")
    return best, pops


def radius_forest(coords_host, r2, rank, device=0):
    """Bottleneck spanning forest of the radius graph (dc_hip_radius_forest).  coords_host: float32
    numpy [n_rows, n_cols]; rank: permutation of 0..n_rows-1 -> (edges uint32 numpy [n_edges, 2] of
    frame ids, number of sweeps)."""
    coords_host = np.ascontiguousarray(coords_host, dtype=np.float32)
    n_rows, n_cols = coords_host.shape
    rank = np.ascontiguousarray(rank, dtype=np.uint32)
    assert rank.shape == (n_rows,)
    edges = np.empty((max(n_rows - 1, 1), 2), dtype=np.uint32)
    n_edges, n_rounds = C.c_size_t(0), C.c_uint32(0)
    rc = capi.lib.dc_hip_radius_forest(coords_host.ctypes.data_as(C.c_void_p), n_rows, n_cols, float(r2),
                                       rank.ctypes.data_as(C.c_void_p), device,
                                       edges.ctypes.data_as(C.c_void_p), C.byref(n_edges),
                                       C.byref(n_rounds))
    capi.check(rc, "dc_hip_radius_forest")
    return edges[:n_edges.value].copy(), int(n_rounds.value)


def pack_neighbors(nn_idx, nn_d2, hd_idx, hd_d2):
    """-> int64 CUDA [2, n_rows]: (d2 bits << 32 | index) words of nn and nn_hd (dc_hip_neighbors_pack_dev);
    partial results of a sharded run merge with all_reduce(min)."""
    n = nn_idx.shape[0]
    words = torch.empty((2, n), dtype=torch.int64, device=nn_idx.device)
    with torch.cuda.device(nn_idx.device):
        rc = capi.lib.dc_hip_neighbors_pack_dev(_dev(nn_idx), _dev(nn_d2), _dev(hd_idx), _dev(hd_d2), n,
                                                _dev(words), _stream_ptr())
    capi.check(rc, "dc_hip_neighbors_pack_dev")
    return words


def unpack_neighbors(words, out=None):
    """inverse of pack_neighbors -> (nn_idx int32, nn_d2 float32, hd_idx int32, hd_d2 float32)"""
    n = words.shape[1]
    dev = words.device
    if out is None:
        out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
               torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
    with torch.cuda.device(dev):
        rc = capi.lib.dc_hip_neighbors_unpack_dev(_dev(words), n, _dev(out[0]), _dev(out[1]), _dev(out[2]),
                                                  _dev(out[3]), _stream_ptr())
    capi.check(rc, "dc_hip_neighbors_unpack_dev")
    return out


def neighbor_block_rows(n_rows, n_cols, n_segments):
    """rows of one rank's block in the all-gather merge of the neighbour partials (dc_hip_neighbors_block_rows)"""
    return int(capi.lib.dc_hip_neighbors_block_rows(n_rows, n_cols, n_segments))


def pack_neighbor_block(coords, nn_idx, nn_d2, hd_idx, hd_d2, segment, n_segments, variant="auto", out=None):
    """The results of this segment's own rows as a dense block int32 [4, block_rows] (dc_hip_neighbors_block_pack_dev);
    call right after nearest_neighbors_segment on the same device (its ordering is read from the workspace)."""
    n_rows, n_cols = _check_coords(coords)
    rows = neighbor_block_rows(n_rows, n_cols, n_segments)
    if out is None:
        out = torch.empty((4, rows), dtype=torch.int32, device=coords.device)
    assert out.shape == (4, rows) and out.dtype == torch.int32 and out.is_contiguous()
    with torch.cuda.device(coords.device):
        ws, ws_bytes = _workspace(coords.device).get(n_rows, n_cols, 1)
        rc = capi.lib.dc_hip_neighbors_block_pack_dev(_dev(nn_idx), _dev(nn_d2), _dev(hd_idx), _dev(hd_d2), n_rows, n_cols,
                                                      segment, n_segments, ws, ws_bytes, capi.VARIANTS[variant],
                                                      _dev(out), _stream_ptr())
    capi.check(rc, "dc_hip_neighbors_block_pack_dev")
    return out


BLOCK_HEADER_ROWS = 32    # the last entries of plane 0 of a block: its layout header (dc_mfma.hip nn_block_pack_kernel)


def check_neighbor_block_layout(blocks):
    """All ranks must have packed their blocks under the SAME layout (by position or row block, the same padded order,
    group size and deal -- words 0..6 of the header); a rank that derived another order would have its rows scattered
    to the wrong frames without any other sign.  One small device comparison + a synchronisation."""
    hdr = blocks[:, 0, -BLOCK_HEADER_ROWS:-BLOCK_HEADER_ROWS + 8]
    if not bool((hdr == hdr[0:1]).all()):
        raise RuntimeError("neighbour blocks of the ranks were packed under different layouts: " + str(hdr.cpu().tolist()))


def layout_status(device):
    """True if the last dc_hip_neighbors_block_unpack_dev in this device's workspace REFUSED its blocks (their layout headers
    differed: nothing was unpacked) -- dc_hip_workspace_layout_status_dev; synchronises"""
    ws = _workspace(device)
    if ws.buf is None:
        return False
    bad = C.c_int(0)
    with torch.cuda.device(device):
        capi.check(capi.lib.dc_hip_workspace_layout_status_dev(_dev(ws.buf), C.byref(bad), _stream_ptr()),
                   "dc_hip_workspace_layout_status_dev")
    return bool(bad.value)


def unpack_neighbor_blocks(coords, blocks, n_segments, variant="auto", out=None, check=True):
    """blocks int32 [n_segments, 4, block_rows] gathered from all ranks -> (nn_idx, nn_d2, hd_idx, hd_d2) by frame
    (dc_hip_neighbors_block_unpack_dev).  The unpack kernel compares the blocks' layout headers itself and writes nothing
    on a mismatch; check=True asks for its verdict right away (one synchronisation) and raises.  check=False leaves that to
    the caller: layout_status() BEFORE the next call into the workspace -- the verdict word describes the last unpack only,
    the next unpack or sweep replaces it -- and on a mismatch the returned arrays hold whatever they held before (zeros
    here: a refused unpack must not hand out uninitialised memory as neighbours)."""
    n_rows, n_cols = _check_coords(coords)
    dev = coords.device
    rows = neighbor_block_rows(n_rows, n_cols, n_segments)
    assert blocks.is_contiguous() and blocks.dtype == torch.int32 and blocks.numel() == n_segments * 4 * rows
    if out is None:
        buf = (torch.empty if check else torch.zeros)((4, n_rows), dtype=torch.int32, device=dev)   # (one fill, not four)
        out = (buf[0], buf[1].view(torch.float32), buf[2], buf[3].view(torch.float32))
    with torch.cuda.device(dev):
        ws, ws_bytes = _workspace(dev).get(n_rows, n_cols, 1)
        if check and not ws_bytes:   # (no workspace for the kernel to flag in -- shapes without a matrix-core sweep: compared here)
            check_neighbor_block_layout(blocks.view(n_segments, 4, rows))
        rc = capi.lib.dc_hip_neighbors_block_unpack_dev(_dev(blocks), n_rows, n_cols, n_segments, ws, ws_bytes,
                                                        capi.VARIANTS[variant], _dev(out[0]), _dev(out[1]), _dev(out[2]),
                                                        _dev(out[3]), _stream_ptr())
    capi.check(rc, "dc_hip_neighbors_block_unpack_dev")
    if check and ws_bytes and layout_status(dev):
        check_neighbor_block_layout(blocks.view(n_segments, 4, rows))   # (raises, with the headers in the message)
        raise RuntimeError("neighbour blocks of the ranks were packed under different layouts")
    return out


def compute_sigma2(nn_d2):
    """compute_sigma2 (density_clustering.cpp:334-343)."""
    out = C.c_double(0.0)
    with torch.cuda.device(nn_d2.device):
        rc = capi.lib.dc_hip_sigma2_dev(_dev(nn_d2), nn_d2.shape[0], C.byref(out), _stream_ptr())
    capi.check(rc, "dc_hip_sigma2_dev")
    return out.value


class Session:
    """A trajectory resident on the GPUs of this process across pop -> FE -> NN -> forest
    (dc_hip_session_*, include/dc_density.h): HOST numpy arrays in and out, one upload per device, partial
    results of several devices merged on the devices over RCCL.  This is the path the C++ shim and the
    command line take; tests use it to compare the resident flow with the call-by-call one."""

    def __init__(self, coords_host, n_devices=0, devices=None):
        c = np.ascontiguousarray(coords_host, dtype=np.float32)
        assert c.ndim == 2
        self.n_rows, self.n_cols = c.shape
        self._h = C.c_void_p(0)
        devs = None
        if devices is not None:
            devs = (C.c_int * len(devices))(*devices)
            n_devices = len(devices)
        capi.check(capi.lib.dc_hip_session_open(c.ctypes.data_as(C.c_void_p), self.n_rows, self.n_cols, devs,
                                                n_devices, C.byref(self._h)), "dc_hip_session_open")

    def close(self):
        if self._h:
            capi.lib.dc_hip_session_close(self._h)
            self._h = C.c_void_p(0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    @property
    def n_devices(self):
        return int(capi.lib.dc_hip_session_devices(self._h))

    @property
    def uses_rccl(self):
        return bool(capi.lib.dc_hip_session_uses_rccl(self._h))

    @property
    def merge_mode(self):
        """0: one device; 1: RCCL collectives on the devices; 2: through the host (the reference's own merge)"""
        return int(capi.lib.dc_hip_session_merge_mode(self._h))

    @property
    def merge_note(self):
        """dc_hip_session_merge_note: which merge runs and, for the host merge, why RCCL is not used"""
        return capi.lib.dc_hip_session_merge_note(self._h).decode("utf-8", "replace")

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.lib.dc_hip_session_counters(self._h, C.byref(a), C.byref(b)), "dc_hip_session_counters")
        return int(a.value), int(b.value)

    def populations(self, radii, fetch=True):
        rad = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
        out = np.empty((rad.size, self.n_rows), dtype=np.uint32) if fetch else None
        capi.check(capi.lib.dc_hip_session_populations(self._h, rad.ctypes.data_as(C.c_void_p), rad.size,
                                                       out.ctypes.data_as(C.c_void_p) if fetch else None),
                   "dc_hip_session_populations")
        return out

    def free_energies(self, radius_index=0, fetch=True):
        out = np.empty(self.n_rows, dtype=np.float32) if fetch else None
        capi.check(capi.lib.dc_hip_session_free_energies(self._h, radius_index,
                                                         out.ctypes.data_as(C.c_void_p) if fetch else None, None),
                   "dc_hip_session_free_energies")
        return out

    def set_free_energies(self, fe):
        f = np.ascontiguousarray(fe, dtype=np.float32)
        assert f.shape == (self.n_rows,)
        capi.check(capi.lib.dc_hip_session_set_free_energies(self._h, f.ctypes.data_as(C.c_void_p)),
                   "dc_hip_session_set_free_energies")

    def nearest_neighbors(self):
        """-> (nn_idx u32, nn_d2 f32, hd_idx u32, hd_d2 f32, sigma2)"""
        n = self.n_rows
        nn_idx, hd_idx = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
        nn_d2, hd_d2 = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.float32)
        s2 = C.c_double(0.0)
        capi.check(capi.lib.dc_hip_session_nearest_neighbors(
            self._h, nn_idx.ctypes.data_as(C.c_void_p), nn_d2.ctypes.data_as(C.c_void_p),
            hd_idx.ctypes.data_as(C.c_void_p), hd_d2.ctypes.data_as(C.c_void_p), C.byref(s2)),
            "dc_hip_session_nearest_neighbors")
        return nn_idx, nn_d2, hd_idx, hd_d2, s2.value

    def radius_forest(self, r2, rank):
        rank = np.ascontiguousarray(rank, dtype=np.uint32)
        assert rank.shape == (self.n_rows,)
        edges = np.empty((max(self.n_rows - 1, 1), 2), dtype=np.uint32)
        n_edges, n_rounds = C.c_size_t(0), C.c_uint32(0)
        capi.check(capi.lib.dc_hip_session_radius_forest(self._h, float(r2), rank.ctypes.data_as(C.c_void_p),
                                                         edges.ctypes.data_as(C.c_void_p), C.byref(n_edges),
                                                         C.byref(n_rounds)), "dc_hip_session_radius_forest")
        return edges[:n_edges.value].copy(), int(n_rounds.value)
