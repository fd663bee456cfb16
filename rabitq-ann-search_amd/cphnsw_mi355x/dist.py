"""Multi-GPU: queries shard across ranks, the index is replicated, results are gathered.

The reference parallelises `search_batch` over queries with OpenMP (src/bindings.cpp:194-212);
queries are independent units, so the MI355X path shards them contiguously across one process
per GPU and needs exactly one collective: the gather of (nq/G) x k ids (int64) and distances
(float32) — 12*k bytes per query, latency-bound on xGMI.  No data-path collective exists.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n queries for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(ids, dist_t, world, group=None, force=False):
    """All-gather equal-sized result shards (torch tensors) along dim 0."""
    if world == 1 and not force:
        return ids, dist_t
    import torch
    import torch.distributed as dist
    out_ids = torch.empty((world * ids.shape[0],) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    out_d = torch.empty((world * dist_t.shape[0],) + tuple(dist_t.shape[1:]), dtype=dist_t.dtype,
                        device=dist_t.device)
    dist.all_gather_into_tensor(out_ids, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(out_d, dist_t.contiguous(), group=group)
    return out_ids, out_d


class PackedResults:
    """One rank's (nq, k) result block as a single byte buffer -- ids (int64) first, distances (float32) behind
    them -- so that the gather is ONE collective of 12*k*nq bytes per rank instead of two.  `ids` and `dist` are views
    into the buffer (pass them as `out=` of `search_batch_device`); `gather()` all-gathers the buffer into a
    preallocated (world, bytes) tensor on the current stream and returns the batch's ids and distances."""

    def __init__(self, nq, k, world, device):
        import torch
        self.nq, self.k, self.world = nq, k, world
        self.buf = torch.empty(nq * k * 12, dtype=torch.uint8, device=device)
        self.ids = self.buf[: nq * k * 8].view(torch.int64).view(nq, k)
        self.dist = self.buf[nq * k * 8:].view(torch.float32).view(nq, k)
        self.all = torch.empty((world, nq * k * 12), dtype=torch.uint8, device=device)

    def gather_raw(self, group=None):
        """The collective alone (what a serving loop enqueues per batch): every rank's buffer into `self.all`."""
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.all.view(-1), self.buf, group=group)
        return self.all

    def gather(self, group=None):
        """gather_raw + the (world*nq, k) ids / distances of the whole batch, rank-major."""
        import torch
        self.gather_raw(group)
        nq, k = self.nq, self.k
        ids = self.all[:, : nq * k * 8].contiguous().view(torch.int64).view(self.world * nq, k)
        d = self.all[:, nq * k * 8:].contiguous().view(torch.float32).view(self.world * nq, k)
        return ids, d


def search_batch_sharded(search_fn, queries, k, group=None, device=None):
    """Runs `search_fn(shard, k) -> (ids, dist)` (numpy) on this rank's shard of `queries`
    (numpy (n, dim), identical on every rank) and returns the full (n, k) result on every rank.
    Ragged shards are padded to the largest shard for the all-gather and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return search_fn(queries, k)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = queries.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    ids, d = search_fn(queries[lo:hi], k)
    biggest = shard_bounds(n, world, 0)[1]
    pad_ids = np.full((biggest, k), -1, np.int64)
    pad_d = np.full((biggest, k), np.finfo(np.float32).max, np.float32)
    pad_ids[: hi - lo] = ids
    pad_d[: hi - lo] = d
    t_ids = torch.from_numpy(pad_ids)
    t_d = torch.from_numpy(pad_d)
    if device is not None:
        t_ids, t_d = t_ids.to(device), t_d.to(device)
    g_ids, g_d = gather_results(t_ids, t_d, world, group)
    g_ids, g_d = g_ids.cpu().numpy().reshape(world, biggest, k), g_d.cpu().numpy().reshape(world, biggest, k)
    out_ids = np.concatenate([g_ids[r, : shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0]]
                              for r in range(world)])
    out_d = np.concatenate([g_d[r, : shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0]]
                            for r in range(world)])
    return out_ids, out_d
