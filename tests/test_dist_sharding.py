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


def _packed_worker(rank, world, port, nq, k, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    import torch
    import torch.distributed as dist
    from golden_util import golden
    from cphnsw_mi355x.dist import PackedResults
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = golden()
        pk = PackedResults(nq, k, world, torch.device("cpu"))
        # each rank fills its views with its shard of the golden result, as search_batch_device(out=...) would
        pk.ids.copy_(torch.from_numpy(g["S/g128/b4/plain/k10/ids"][rank * nq:(rank + 1) * nq]))
        pk.dist.copy_(torch.from_numpy(g["S/g128/b4/plain/k10/d"][rank * nq:(rank + 1) * nq]))
        ids, d = pk.gather()
        q.put((rank, ids.numpy(), d.numpy()))
    finally:
        dist.destroy_process_group()


def test_packed_gather_is_one_collective_and_rank_major(gold):
    """PackedResults (bench.py's N > 1 step): ids and distances travel in one byte buffer per rank."""
    world, k, nq = 2, 10, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, nq, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ids, d in res:
        assert np.array_equal(ids, gold["S/g128/b4/plain/k10/ids"][:world * nq])
        assert d.tobytes() == gold["S/g128/b4/plain/k10/d"][:world * nq].tobytes()


def _bench_step_worker(rank, world, port, nq, k, q):
    """bench.py's own N > 1 control flow (make_step + timed_region: priming, warm-up, barrier, K steps alternating two
    scratch sets, the packed all-gather inside the step, MAX of the elapsed time over ranks) on CPU tensors over gloo.
    The search itself is stood in for by the oracle on the rank's query shard (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    import bench
    from golden_util import fixture_path, golden
    from oracle_lib import Oracle
    from cphnsw_mi355x.dist import PackedResults
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oi = Oracle().load(fixture_path("g128", 4))
        Q = golden()["Q/g128"]
        q_shard = torch.from_numpy(Q[rank * nq:(rank + 1) * nq])
        calls = []

        def search_device(qs, kk, out, stream):
            ids, d, _ = oi.search_batch(qs.numpy(), kk, nthreads=1)
            out[0].copy_(torch.from_numpy(ids))
            out[1].copy_(torch.from_numpy(d))
            calls.append(stream)
            if rank == 1:
                time.sleep(0.01)          # an uneven rank: the reported time must be the slowest rank's
            return out

        dev = torch.device("cpu")
        packs = [PackedResults(nq, k, world, dev) for _ in range(2)]
        step = bench.make_step(search_device, q_shard, k, packs, ["s0", "s1"], True, lambda st: bench.NullStream())
        steps, warmup = 4, 1
        t0 = time.perf_counter()
        el, ids, d = bench.timed_region(step, steps, warmup, False, True, dist, lambda: None, dev)
        wall = time.perf_counter() - t0
        assert len(calls) == 4 + warmup + steps and calls[-2:] == ["s0", "s1"]       # steps alternate the two sets
        # after the last step (i = 3 -> set 1) that set's gather buffer holds the whole batch, rank-major
        nqk = nq * k
        all_ids = packs[1].all[:, : nqk * 8].contiguous().view(torch.int64).view(world * nq, k)
        all_d = packs[1].all[:, nqk * 8:].contiguous().view(torch.float32).view(world * nq, k)
        q.put((rank, el, wall, all_ids.numpy().copy(), all_d.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_bench_step_and_timed_region_over_gloo(gold):
    """The N > 1 path of bench.py executed end to end on two ranks before the driver's first multi-GPU run."""
    world, k, nq = 2, 10, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_step_worker, args=(r, world, port, nq, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    els = {r[0]: r[1] for r in res}
    assert els[0] == els[1]                                  # the all-reduced MAX: every rank reports the same time
    assert els[0] >= 4 * 0.01                                # ... the slow rank's (4 timed steps x 10 ms of extra work)
    for _, el, wall, ids, d in res:
        assert el <= wall
        assert np.array_equal(ids, gold["S/g128/b4/plain/k10/ids"][:world * nq])
        assert d.tobytes() == gold["S/g128/b4/plain/k10/d"][:world * nq].tobytes()


def test_shard_bounds_cover_everything():
    from cphnsw_mi355x.dist import shard_bounds
    for n in (0, 1, 7, 8, 10_000, 10_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _gpu_worker(port, q):
    """One rank over RCCL: the HIP search_batch as the shard's search function, device tensors gathered."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
    import torch
    import torch.distributed as dist
    from golden_util import fixture_path, golden
    import cphnsw_mi355x
    from cphnsw_mi355x.dist import PackedResults, gather_results, search_batch_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ix = cphnsw_mi355x.CPIndex(128, 4, device=0)
        ix.load(fixture_path("g128", 4))
        Q = golden()["Q/g128"]
        ids, d = search_batch_sharded(ix.search_batch, Q, 10, device=dev)      # numpy in/out through the all-gather
        # the device-resident path of bench.py: search_batch_device on a side stream + all_gather_into_tensor
        st = torch.cuda.Stream(dev)
        qd = torch.from_numpy(Q).to(dev)
        torch.cuda.synchronize()
        di, dd = ix.search_batch_device(qd, 10, stream=st)
        with torch.cuda.stream(st):
            gi, gd = gather_results(di, dd, 1, force=True)
        st.synchronize()
        # ... and exactly bench.py's step: results written into the packed buffer, one collective behind the search
        pk = PackedResults(len(Q), 10, 1, dev)
        ix.search_batch_device(qd, 10, out=(pk.ids, pk.dist), stream=st)
        with torch.cuda.stream(st):
            pi, pd = pk.gather()
        st.synchronize()
        q.put((ids, d, gi.cpu().numpy(), gd.cpu().numpy(), pi.cpu().numpy(), pd.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_gather_of_hip_search_matches_goldens(gold):
    """search_batch_sharded / gather_results with the HIP search over RCCL (nccl backend, world_size 1 on the
    one-GPU box): the collective path bench.py times for N > 1, against the reference's goldens."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_gpu_worker, args=(_free_port(), q))
    p.start()
    ids, d, gi, gd, pi, pd = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    for a, b in ((ids, d), (gi, gd), (pi, pd)):
        assert np.array_equal(a, gold["S/g128/b4/plain/k10/ids"])
        assert b.tobytes() == gold["S/g128/b4/plain/k10/d"].tobytes()


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must become a 2-rank run by itself (the parent starts
    `python -m torch.distributed.run` before it touches torch / the GPU and relays rank 0's line).  Driven here through
    bench.py's test hook: CPU tensors over gloo, a stand-in for the search, the real launcher / sharding / step / gather /
    timed-region code."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--stub-search"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["stub"] is True
    assert j["rows"] == 96                    # two shards of 48 queries, gathered rank-major inside the step
    # the gathered ids are what one process computes for the whole batch
    Q = np.random.default_rng(7).standard_normal((96, 16)).astype(np.float32)
    key = (np.abs(Q).sum(axis=1, dtype=np.float32) * np.float32(1000.0)).astype(np.int64)
    assert j["ids_checksum"] == int((key[:, None] + np.arange(10)[None, :]).sum())
    assert j["per_rank_qps"]["min"] <= j["per_rank_qps"]["max"] and j["value"] <= 2 * j["per_rank_qps"]["max"] * 1.001
    # a launcher that disagrees with --gpus is an error, not a silently mislabelled run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-search"],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="3"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
