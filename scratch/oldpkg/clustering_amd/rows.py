"""Row partition shared by the single-process and the distributed host code."""


def shard_rows(n_rows, n_shards, shard):
    """Row block of one device, exactly as density_clustering_cuda.cu:149,165-169:
    floor(N/G) rows each, the last device takes the remainder."""
    rng = n_rows // n_shards
    lo = shard * rng
    hi = n_rows if shard == n_shards - 1 else (shard + 1) * rng
    return lo, hi
