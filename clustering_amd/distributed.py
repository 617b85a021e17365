"""Row-sharded density path over torch.distributed (one process per GPU; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" for the CPU tests of this host logic).

Sharding and merging follow the reference's multi-GPU host code:
  - contiguous row blocks [g*floor(N/G), (g+1)*floor(N/G)), the last rank takes the remainder
    (density_clustering_cuda.cu:149,165-169 / :293,305-308); coordinates are replicated;
  - populations: every rank holds [n_radii][N], zero outside its rows; the reference sums the
    partials on the host (density_clustering_cuda.cu:171-180) -> here ONE all-reduce(sum) of
    n_radii*N int32 (exact);
  - free energies: every rank computes all N values locally from the reduced populations;
  - neighbours: the reference overwrites row by row from each partial (cuda.cu:311-326) -> here
    ONE all-gather of per-rank blocks [4][block] (nn_idx, nn_d2 bits, hd_idx, hd_d2 bits), blocks
    padded to the largest (= last) shard.

Neighbour merge of the segments (default since round 3): an ALL-GATHER of position-ordered blocks.  Every
rank compacts the results of its own segment's rows into a dense block [4][block_rows] (by local position in the
sweep's spatial order, which all ranks derive identically from the replicated inputs), one
all_gather_into_tensor moves the blocks, one kernel scatters them back to frame order: half the bytes of the
all-reduce(min) below and no reduction.  DC_NN_MERGE=allreduce selects the older merge.

Spatial segments.  A block of consecutive trajectory rows is spread over the whole conformational
space, so a rank's query groups are far less compact than those of a full sweep and the tile-pair
pruning loses a third of its effect (measured: 29 % instead of 20 % of the tile pairs at 1/8 of C3).
A backend that offers ``*_segment`` methods (the HIP backend does) is therefore asked for SEGMENT
g of G instead: every G-th query group of the sweep's spatial order (groups are dealt out cyclically, so
every rank gets the same mix of dense and sparse regions).  Populations merge exactly
as before (partial counts, all-reduce(sum)); the neighbour rows of a segment are scattered
over the trajectory, so they merge by ONE all-reduce(min) of [2][N] int64 words
(d2 bits << 32 | index): every row has exactly one owner, and the "none" value (N+1, FLT_MAX) that
all the other ranks hold for it is larger than anything the owner can report.
"""
import torch
import torch.distributed as dist

from .rows import shard_rows


class HipBackend:
    """The product compute backend: libdcdensity.so through clustering_amd.density (no fallback)."""

    def __init__(self, variant="auto"):
        from . import density
        self._d = density
        self.variant = variant

    def populations_partial(self, coords, radii, lo, hi):
        return self._d.calculate_populations_partial(coords, radii, lo, hi, variant=self.variant)

    def free_energies(self, pops_row):
        return self._d.calculate_free_energies(pops_row)

    def nearest_neighbors_partial(self, coords, fe, lo, hi, stats_valid=False):
        return self._d.nearest_neighbors_partial(coords, fe, lo, hi, variant=self.variant, stats_valid=stats_valid)

    def populations_segment(self, coords, radii, segment, n_segments):
        return self._d.calculate_populations_segment(coords, radii, segment, n_segments, variant=self.variant)

    def nearest_neighbors_segment(self, coords, fe, segment, n_segments, stats_valid=False):
        return self._d.nearest_neighbors_segment(coords, fe, segment, n_segments, variant=self.variant,
                                                 stats_valid=stats_valid)

    accepts_stats_valid = True      # ShardedDensity may tell the neighbour call that the header statistics are valid

    def pack_neighbor_block(self, coords, nn, segment, n_segments):
        return self._d.pack_neighbor_block(coords, nn[0], nn[1], nn[2], nn[3], segment, n_segments, variant=self.variant)

    def unpack_neighbor_blocks(self, coords, blocks, n_segments, check=True):
        return self._d.unpack_neighbor_blocks(coords, blocks, n_segments, variant=self.variant, check=check)

    def layout_status(self, device):
        return self._d.layout_status(device)

    def pack_neighbors(self, nn_idx, nn_d2, hd_idx, hd_d2):
        return self._d.pack_neighbors(nn_idx, nn_d2, hd_idx, hd_d2)

    def unpack_neighbors(self, words):
        return self._d.unpack_neighbors(words)

    def radius_min_edge_segment(self, coords, r2, comp, rank, segment, n_segments):
        return self._d.radius_min_edge(coords, r2, comp, rank, segment, n_segments)[0]


class ShardedDensity:
    """pop -> FE -> NN for the rows of this rank, merged across ranks with two collectives."""

    def __init__(self, backend=None, group=None, check_layout=True):
        """check_layout: ask the unpack kernel for its verdict on the gathered blocks' layout headers in every step (one
        synchronisation).  False: the caller asks with check_layouts() after a step and BEFORE the next one -- the verdict
        word is that of the last unpack only, the next step's sweeps and unpack replace it; a refused unpack writes
        nothing (the step's neighbour arrays are zeros then)"""
        self.backend = backend if backend is not None else HipBackend()
        self.group = group
        self.check_layout = check_layout

    def check_layouts(self, device):
        """raises if the LAST unpack on this device refused its blocks (ranks derived different orders); earlier steps are
        not covered (dc_hip_workspace_layout_status_dev)"""
        if hasattr(self.backend, "layout_status") and self.backend.layout_status(device):
            raise RuntimeError("neighbour blocks of the ranks were packed under different layouts")

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def neighbour_merge(self):
        """"allgather" (position-ordered blocks; needs a backend with pack_neighbor_block) or "allreduce" (packed
        (d2, index) words, minimum); DC_NN_MERGE overrides"""
        import os
        want = os.environ.get("DC_NN_MERGE", "allgather")
        if want == "allgather" and hasattr(self.backend, "pack_neighbor_block"):
            return "allgather"
        return "allreduce"

    def neighbour_merge_name(self):
        if self.neighbour_merge() == "allgather":
            return "all-gather of position-ordered neighbour blocks [4][block_rows] per rank"
        return "all-reduce(min) of the packed (d2, index) neighbour words"

    def run(self, coords, radii, fe_radius_index=0, want_nn=True, mark=None):
        """coords: [N, D] float32 on this rank's device (replicated).  Returns a dict of tensors on
        that device: pops int32 [n_radii, N], fe float32 [N], and if want_nn nn_idx/hd_idx int32 [N],
        nn_d2/hd_d2 float32 [N] -- identical on every rank.  mark(name), if given, is called at the start and
        after every phase: "start", "pop", "pops_allreduce", "fe", "nn", "nn_merge" (bench.py records events)."""
        mark = mark or (lambda name: None)
        rank, world = self._world()
        n_rows = coords.shape[0]
        lo, hi = shard_rows(n_rows, world, rank)
        segments = world > 1 and hasattr(self.backend, "populations_segment")
        mark("start")
        if segments:
            pops = self.backend.populations_segment(coords, radii, rank, world)
        else:
            pops = self.backend.populations_partial(coords, radii, lo, hi)
        mark("pop")
        if world > 1:
            dist.all_reduce(pops, op=dist.ReduceOp.SUM, group=self.group)
        mark("pops_allreduce")
        fe = self.backend.free_energies(pops[fe_radius_index].contiguous())
        mark("fe")
        out = {"pops": pops, "fe": fe}
        if not want_nn:
            return out
        # the neighbour sweep runs over the coordinates the population sweep just went through: its statistics
        # passes (column means, max norm, bounding box) are skipped (DC_FLAG_STATS_VALID)
        kw = {"stats_valid": True} if getattr(self.backend, "accepts_stats_valid", False) else {}
        if segments:
            nn = self.backend.nearest_neighbors_segment(coords, fe, rank, world, **kw)
            mark("nn")
            if self.neighbour_merge() == "allgather":
                block = self.backend.pack_neighbor_block(coords, nn, rank, world)
                gathered = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
                dist.all_gather_into_tensor(gathered.view(-1), block.view(-1), group=self.group)
                if hasattr(self.backend, "layout_status"):
                    nn_idx, nn_d2, hd_idx, hd_d2 = self.backend.unpack_neighbor_blocks(coords, gathered, world, check=self.check_layout)
                else:
                    nn_idx, nn_d2, hd_idx, hd_d2 = self.backend.unpack_neighbor_blocks(coords, gathered, world)
            elif hasattr(self.backend, "pack_neighbors"):     # two library kernels instead of a dozen torch ops
                # (d2 bits << 32 | index): d2 >= 0, so the words order like (d2, index); one owner per row
                packed = self.backend.pack_neighbors(*nn)
                dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=self.group)
                nn_idx, nn_d2, hd_idx, hd_d2 = self.backend.unpack_neighbors(packed)
            else:
                nn_idx, nn_d2, hd_idx, hd_d2 = nn
                packed = torch.empty((2, n_rows), dtype=torch.int64, device=coords.device)
                packed[0] = (nn_d2.view(torch.int32).to(torch.int64) << 32) | (nn_idx.to(torch.int64) & 0xFFFFFFFF)
                packed[1] = (hd_d2.view(torch.int32).to(torch.int64) << 32) | (hd_idx.to(torch.int64) & 0xFFFFFFFF)
                dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=self.group)
                nn_idx = (packed[0] & 0xFFFFFFFF).to(torch.int32)
                hd_idx = (packed[1] & 0xFFFFFFFF).to(torch.int32)
                nn_d2 = (packed[0] >> 32).to(torch.int32).view(torch.float32)
                hd_d2 = (packed[1] >> 32).to(torch.int32).view(torch.float32)
            mark("nn_merge")
            out.update(nn_idx=nn_idx, nn_d2=nn_d2, hd_idx=hd_idx, hd_d2=hd_d2)
            return out
        nn_idx, nn_d2, hd_idx, hd_d2 = self.backend.nearest_neighbors_partial(coords, fe, lo, hi, **kw)
        mark("nn")
        if world > 1:
            block = n_rows - (world - 1) * (n_rows // world)        # largest shard (the last one)
            send = torch.zeros((4, block), dtype=torch.int32, device=coords.device)
            send[0, :hi - lo] = nn_idx[lo:hi]
            send[1, :hi - lo] = nn_d2[lo:hi].view(torch.int32)
            send[2, :hi - lo] = hd_idx[lo:hi]
            send[3, :hi - lo] = hd_d2[lo:hi].view(torch.int32)
            recv = torch.empty((world, 4, block), dtype=torch.int32, device=coords.device)
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
            for g in range(world):
                glo, ghi = shard_rows(n_rows, world, g)
                nn_idx[glo:ghi] = recv[g, 0, :ghi - glo]
                nn_d2[glo:ghi] = recv[g, 1, :ghi - glo].view(torch.float32)
                hd_idx[glo:ghi] = recv[g, 2, :ghi - glo]
                hd_d2[glo:ghi] = recv[g, 3, :ghi - glo].view(torch.float32)
        mark("nn_merge")
        out.update(nn_idx=nn_idx, nn_d2=nn_d2, hd_idx=hd_idx, hd_d2=hd_d2)
        return out


class ShardedForest:
    """Bottleneck spanning forest of the radius graph (the screening of a -T scan that starts from an empty
    clustering) with the Boruvka rounds sharded over the ranks: every rank sweeps the
    query groups of its segment (dc_hip_radius_min_edge_segment_dev), the per-component candidates merge with
    ONE all-reduce(min) of n_rows int64 words per round, and every rank merges the components itself (same
    input, same result -- no broadcast)."""

    NONE = (1 << 63) - 1   # "no pair leaves this component" on the wire (the library's ~0 is -1 as int64)

    def __init__(self, backend=None, group=None):
        self.backend = backend if backend is not None else HipBackend()
        self.group = group

    def run(self, coords, r2, rank):
        """coords [N, D] float32 (replicated), rank int32 [N]: a permutation of 0..N-1 (position in order of
        free energy) on the same device -> (edges int64 numpy [n_edges, 2] of frame ids, rounds)."""
        import numpy as np
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        if dist.is_available() and dist.is_initialized():
            me, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        else:
            me, world = 0, 1
        n = coords.shape[0]
        rank_h = rank.cpu().numpy().astype(np.int64)
        frame_of = np.empty(n, dtype=np.int64)
        frame_of[rank_h] = np.arange(n)
        comp_h = np.arange(n, dtype=np.int64)            # id of a component = its smallest frame id
        edges = []
        rounds = 0
        while n > 1 and rounds < 64:
            rounds += 1
            comp = torch.from_numpy(comp_h.astype(np.int32)).to(coords.device)
            best = self.backend.radius_min_edge_segment(coords, r2, comp, rank, me, world if world > 1 else 0)
            best = torch.where(best < 0, torch.full_like(best, self.NONE), best)
            if world > 1:
                dist.all_reduce(best, op=dist.ReduceOp.MIN, group=self.group)
            b = best.cpu().numpy()
            picked = np.nonzero(b != self.NONE)[0]
            if picked.size == 0:
                break
            a_f, b_f = frame_of[b[picked] >> 32], frame_of[b[picked] & 0xFFFFFFFF]
            pairs = np.unique(np.stack([np.minimum(a_f, b_f), np.maximum(a_f, b_f)], axis=1), axis=0)
            # distinct weights: the picks of one round never close a cycle, every distinct pair is a forest edge
            edges.append(pairs)
            g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (comp_h[pairs[:, 0]], comp_h[pairs[:, 1]])),
                           shape=(n, n))
            _, lab = connected_components(g, directed=False)
            lab_of_frame = lab[comp_h]
            smallest = np.full(lab.max() + 1, n, dtype=np.int64)
            np.minimum.at(smallest, lab_of_frame, np.arange(n))
            comp_h = smallest[lab_of_frame]
        out = np.concatenate(edges) if edges else np.zeros((0, 2), dtype=np.int64)
        return out, rounds
