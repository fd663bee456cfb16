"""Multi-rank query sharding on CPU: world_size 2 over gloo.  The search function is the oracle
(test infrastructure) — what is under test is the shard/gather logic of cphnsw_mi355x.dist,
which must reproduce the single-process result exactly, including ragged shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, k, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    import torch.distributed as dist
    from golden_util import fixture_path, golden
    from oracle_lib import Oracle
    from cphnsw_mi355x.dist import search_batch_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oi = Oracle().load(fixture_path("g128", 4))
        Q = golden()["Q/g128"][:nq]

        def fn(shard, kk):
            if len(shard) == 0:
                return np.zeros((0, kk), np.int64), np.zeros((0, kk), np.float32)
            ids, d, _ = oi.search_batch(shard, kk, nthreads=1)
            return ids, d

        ids, d = search_batch_sharded(fn, Q, k)
        q.put((rank, ids, d))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nq", [24, 7, 1])
def test_two_rank_sharding_matches_single_process(gold, nq):
    world, k = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nq, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_ids = gold["S/g128/b4/plain/k10/ids"][:nq]
    want_d = gold["S/g128/b4/plain/k10/d"][:nq]
    for _, ids, d in res:
        assert np.array_equal(ids, want_ids)
        assert d.tobytes() == want_d.tobytes()


def test_shard_bounds_cover_everything():
    from cphnsw_mi355x.dist import shard_bounds
    for n in (0, 1, 7, 8, 10_000, 10_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
