"""CPU, world_size 2, gloo: the row-sharding + merge logic of clustering_amd.distributed
(all-reduce of zero-padded populations, all-gather of padded neighbour blocks) -- the compute
backend is injected (the CPU oracle's per-row-range functions stand in for the GPU kernels, which
cannot run here); the product's own backend (HipBackend) is what bench.py / the GPU tests use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clustering_amd.rows import shard_rows
from clustering_amd.synth import gaussian_blobs


class OracleBackend:
    def __init__(self):
        from oracle.oracle import Oracle
        self.o = Oracle()

    def populations_partial(self, coords, radii, lo, hi):
        p = self.o.populations(coords.numpy(), radii, lo, hi)
        return torch.from_numpy(p.astype(np.int32))

    def free_energies(self, pops_row):
        return torch.from_numpy(self.o.free_energies(pops_row.numpy().astype(np.uint64)))

    def nearest_neighbors_partial(self, coords, fe, lo, hi):
        a, b, c, d = self.o.nearest_neighbors(coords.numpy(), fe.numpy(), lo, hi)
        return (torch.from_numpy(a.astype(np.int32)), torch.from_numpy(b),
                torch.from_numpy(c.astype(np.int32)), torch.from_numpy(d))


class SegmentOracleBackend(OracleBackend):
    """Offers the segment interface of the HIP backend: segment g of G owns the rows i with i % G == g
    (scattered on purpose, like the spatial segments of the pruned sweep), so that ShardedDensity takes
    its all-reduce(min) merge path."""

    def populations_segment(self, coords, radii, segment, n_segments):
        p = self.populations_partial(coords, radii, 0, coords.shape[0])
        mask = (torch.arange(coords.shape[0]) % n_segments) == segment
        return p * mask.to(p.dtype)

    def nearest_neighbors_segment(self, coords, fe, segment, n_segments):
        n = coords.shape[0]
        a, b, c, d = self.nearest_neighbors_partial(coords, fe, 0, n)
        mask = (torch.arange(n) % n_segments) == segment
        none_i = torch.full_like(a, n + 1)
        none_d = torch.full_like(b, torch.finfo(torch.float32).max)
        return (torch.where(mask, a, none_i), torch.where(mask, b, none_d),
                torch.where(mask, c, none_i), torch.where(mask, d, none_d))


class BlockSegmentOracleBackend(SegmentOracleBackend):
    """... and the block interface of the HIP backend, so that ShardedDensity takes its all-gather merge: the rows of
    segment g compacted into a dense block [4][ceil(N/G)] by LOCAL position (local l of segment g = frame l*G + g --
    the stand-in for "position in the sweep's spatial order"), gathered, scattered back to frame order."""

    @staticmethod
    def _rows(n, n_segments):
        return (n + n_segments - 1) // n_segments

    def pack_neighbor_block(self, coords, nn, segment, n_segments):
        n = coords.shape[0]
        rows = self._rows(n, n_segments)
        block = torch.empty((4, rows), dtype=torch.int32)
        block[0::2] = n + 1
        block[1::2] = torch.tensor(np.float32(np.finfo(np.float32).max)).view(torch.int32)
        own = torch.arange(segment, n, n_segments)
        for c, t in enumerate(nn):
            block[c, :own.numel()] = t[own].view(torch.int32)
        return block

    def unpack_neighbor_blocks(self, coords, blocks, n_segments):
        n = coords.shape[0]
        frames = torch.arange(n)
        g, l = frames % n_segments, frames // n_segments
        cols = [blocks[g, c, l] for c in range(4)]
        return cols[0], cols[1].view(torch.float32), cols[2], cols[3].view(torch.float32)


class MinEdgeOracleBackend:
    """dc_hip_radius_min_edge_segment_dev restated with the oracle's pairwise d2: segment g of G sees the
    pairs from the query rows i with i % G == g."""

    def __init__(self):
        from oracle.oracle import Probe, build
        build()
        self.p = Probe()
        self._d2 = None

    def radius_min_edge_segment(self, coords, r2, comp, rank, segment, n_segments):
        c = coords.numpy()
        n = c.shape[0]
        if self._d2 is None:
            self._d2 = self.p.pairwise_d2(c)
        comp_h, rank_h = comp.numpy().astype(np.int64), rank.numpy().astype(np.int64)
        best = np.full(n, -1, dtype=np.int64).view(np.uint64)
        g = max(n_segments, 1)
        for i in range(n):
            if i % g != (segment if n_segments else 0):
                continue
            js = np.nonzero((self._d2[i] < np.float32(r2)) & (comp_h != comp_h[i]))[0]
            if js.size:
                keys = (np.maximum(rank_h[i], rank_h[js]).astype(np.uint64) << np.uint64(32)) | \
                    np.minimum(rank_h[i], rank_h[js]).astype(np.uint64)
                best[comp_h[i]] = min(best[comp_h[i]], keys.min())
        return torch.from_numpy(best.view(np.int64).copy())


def _forest_worker(rank, world, port, n_rows, r2, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clustering_amd.distributed import ShardedForest
        coords = torch.from_numpy(gaussian_blobs(n_rows, 3, seed=77))
        fe_rank = torch.from_numpy(np.random.default_rng(5).permutation(n_rows).astype(np.int32))
        edges, rounds = ShardedForest(MinEdgeOracleBackend()).run(coords, r2, fe_rank)
        np.savez(os.path.join(out_dir, f"forest{rank}.npz"), edges=edges, rounds=rounds)
    finally:
        dist.destroy_process_group()


def _components(n, pairs):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in pairs:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return np.array([find(i) for i in range(n)])


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_forest_has_the_connectivity_of_the_radius_graph(tmp_path, probe, world):
    """ShardedForest: Boruvka rounds with the candidates of the segments merged by all-reduce(min) -- a forest
    of pairs of the radius graph with its connectivity below every rank threshold, identical on all ranks"""
    n_rows, r2 = 400, 0.01
    mp.spawn(_forest_worker, args=(world, _free_port(), n_rows, r2, str(tmp_path)), nprocs=world, join=True)
    c = gaussian_blobs(n_rows, 3, seed=77)
    d2 = probe.pairwise_d2(c)
    fe_rank = np.random.default_rng(5).permutation(n_rows).astype(np.int64)
    ii, jj = np.nonzero(np.triu(d2 < np.float32(r2), k=1))
    all_pairs = np.stack([ii, jj], axis=1)
    first = np.load(os.path.join(tmp_path, "forest0.npz"))["edges"]
    for rank in range(world):
        assert (np.load(os.path.join(tmp_path, f"forest{rank}.npz"))["edges"] == first).all()
    pair_set = {(int(a), int(b)) for a, b in all_pairs}
    assert all((int(min(a, b)), int(max(a, b))) in pair_set for a, b in first)
    assert len(first) == n_rows - len(np.unique(_components(n_rows, all_pairs))), "not a spanning forest"
    w_all = np.maximum(fe_rank[all_pairs[:, 0]], fe_rank[all_pairs[:, 1]])
    w_for = np.maximum(fe_rank[first[:, 0]], fe_rank[first[:, 1]])
    for t in [0, 60, 150, 250, 400]:
        assert (_components(n_rows, all_pairs[w_all < t]) == _components(n_rows, first[w_for < t])).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_rows, out_dir, segments=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clustering_amd.distributed import ShardedDensity
        coords = torch.from_numpy(gaussian_blobs(n_rows, 5, seed=99))
        backend = {False: OracleBackend, True: SegmentOracleBackend, "blocks": BlockSegmentOracleBackend}[segments]()
        job = ShardedDensity(backend)
        if world > 1:
            assert job.neighbour_merge() == ("allgather" if segments == "blocks" else "allreduce")
        phases = []
        out = job.run(coords, [0.1, 0.2], fe_radius_index=1, want_nn=True, mark=phases.append)
        assert phases == ["start", "pop", "pops_allreduce", "fe", "nn", "nn_merge"]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: v.numpy() for k, v in out.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows,segments", [(2, 1001, False), (2, 64, False), (3, 500, False),
                                                   (2, 777, True), (3, 500, True), (2, 777, "blocks"),
                                                   (3, 500, "blocks"), (3, 2, "blocks")])
def test_sharded_density_matches_single_process(tmp_path, oracle, world, n_rows, segments):
    """row blocks + all-gather (segments=False), scattered segments + all-reduce(min) of the packed (d2, index)
    words (segments=True) and scattered segments + all-gather of position-ordered blocks (segments="blocks": the
    path the HIP backend takes)"""
    mp.spawn(_worker, args=(world, _free_port(), n_rows, str(tmp_path), segments), nprocs=world, join=True)
    c = gaussian_blobs(n_rows, 5, seed=99)
    pops = oracle.populations(c, [0.1, 0.2])
    fe = oracle.free_energies(pops[1])
    nn = oracle.nearest_neighbors(c, fe)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert (got["pops"].astype(np.uint64) == pops).all()
        assert (got["fe"].view(np.uint32) == fe.view(np.uint32)).all()
        assert (got["nn_idx"].astype(np.uint64) == nn[0]).all()
        assert (got["hd_idx"].astype(np.uint64) == nn[2]).all()
        assert (got["nn_d2"].view(np.uint32) == nn[1].view(np.uint32)).all()
        assert (got["hd_d2"].view(np.uint32) == nn[3].view(np.uint32)).all()


def test_shard_rows_is_the_reference_partition():
    # density_clustering_cuda.cu:149,165-169: floor(N/G) each, last takes the remainder
    assert [shard_rows(10, 3, g) for g in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [shard_rows(8, 8, g) for g in range(8)] == [(g, g + 1) for g in range(8)]
    assert shard_rows(5, 1, 0) == (0, 5)
    assert [shard_rows(2, 4, g) for g in range(4)] == [(0, 0), (0, 0), (0, 0), (0, 2)]
