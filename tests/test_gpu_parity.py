"""GPU parity: the HIP path (through the C-ABI / Python drop-in) against the reference's golden
vectors and the oracle.  Bit-exact everywhere: integer sums, fp32 estimates and bounds
(tolerance stated by north_star is 1e-4 relative — we hold 0 ulp), exact-L2 distances and
the final ids/distances including duplicates and ties.
"""
import os

import numpy as np
import pytest

from golden_util import DATASETS, KS, fixture_path

pytestmark = pytest.mark.gpu

CASES = [(n, b, v) for n, s in DATASETS.items() for b in s["bits"] for v in s["variants"]]


def _beq(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.fixture(scope="module")
def cph():
    import cphnsw_mi355x
    return cphnsw_mi355x


def _load(cph, name, bits, variant="plain"):
    ix = cph.CPIndex(DATASETS[name]["dim"], bits)
    ix.load(fixture_path(name, bits, variant))
    return ix


@pytest.mark.parametrize("name,bits,variant", CASES)
def test_search_batch_matches_reference(cph, gold, name, bits, variant):
    ix = _load(cph, name, bits, variant)
    Q = gold[f"Q/{name}"]
    for k in KS:
        ids, d = ix.search_batch(Q, k)
        assert ids.dtype == np.int64 and d.dtype == np.float32 and ids.shape == (len(Q), k)
        assert np.array_equal(ids, gold[f"S/{name}/b{bits}/{variant}/k{k}/ids"]), (name, bits, variant, k)
        assert _beq(d, gold[f"S/{name}/b{bits}/{variant}/k{k}/d"]), (name, bits, variant, k)
    # The 24 queries above took the small-batch launch (full-capacity slots, the instantiation that fetches whole blocks).
    # Explicit search parameters keep the general path: 8 slots and a queue, the launch order, and for 4-bit codes at
    # D = 128 the probe-first instantiation followed by the re-run launch -- `bignop` makes every third vertex' stage-1
    # bounds fail, so queries whose stage-2 decision needs the unfetched codes are handed over there.
    ix.set_search_params(slots=8, beam_capacity=0)
    for k in (10, 100):
        ids, d = ix.search_batch(Q, k)
        assert np.array_equal(ids, gold[f"S/{name}/b{bits}/{variant}/k{k}/ids"]), (name, bits, variant, k, "general path")
        assert _beq(d, gold[f"S/{name}/b{bits}/{variant}/k{k}/d"]), (name, bits, variant, k, "general path")
    st = ix.last_search_stats()
    if variant == "bignop" and bits == 4:
        print("bignop general path:", st)


@pytest.mark.parametrize("name,bits,variant", [c for c in CASES if c[2] in ("plain", "gamma")])
def test_search_single_unpadded(cph, gold, name, bits, variant):
    ix = _load(cph, name, bits, variant)
    Q = gold[f"Q/{name}"]
    for qi in range(4):
        ids, d = ix.search(Q[qi], 10)
        assert np.array_equal(ids, gold[f"S1/{name}/b{bits}/{variant}/q{qi}/ids"])
        assert _beq(d, gold[f"S1/{name}/b{bits}/{variant}/q{qi}/d"])
    # float64 queries are force-cast like pybind11's forcecast
    ids64, _ = ix.search(Q[1].astype(np.float64), 10)
    assert np.array_equal(ids64, gold[f"S1/{name}/b{bits}/{variant}/q1/ids"])


@pytest.mark.parametrize("name,bits,variant", [(n, b, v) for n, s in DATASETS.items() for b in s["bits"]
                                               for v in ("plain", "shortcount") if v in s["variants"]])
def test_fastscan_block_and_exact_l2(cph, oracle, gold, name, bits, variant):
    """`shortcount`: every third vertex' list is cut to 0 / 5 / 13 / 29 / 31 entries -- the lanes from 8 * (count / 8) on take
    the reference's scalar tail (different FMA contraction, tests/golden F/../c<count>), the lanes behind count are unused."""
    ix = _load(cph, name, bits, variant)
    oi = oracle.load(fixture_path(name, bits, variant))
    Q = gold[f"Q/{name}"]
    rng = np.random.default_rng(5)
    D = DATASETS[name]["D"]
    for qi in (0, 3, 7):
        lut, co = ix.encode_query(Q[qi])
        olut, oco, _ = oracle.encode_query(Q[qi], D)
        assert np.array_equal(lut, olut) and _beq(co, oco)
        assert ix.entry_point(Q[qi]) == oi.entry_point(Q[qi])
        if qi == 0:   # the upper-layer descent (a12) for every query of the fixture, not just these three
            for q in Q:
                assert ix.entry_point(q) == oi.entry_point(q)
        for qp_tail, dqp in (((1.0, 0.0, 0.0, 0.1), 37.5), ((0.93, 0.02, 0.55, -0.05), 2.5e3),
                             ((1.0, 0.0, 0.0, 0.0), 0.0), ((1.0, 0.0, 0.0, 0.3), 5e-13)):
            qp = np.array([co[0], co[1], co[2], *qp_tail], np.float32)
            verts = list(rng.integers(0, oi.n, 6))
            if variant == "shortcount":
                verts += [1, 4, 7, 10, 13]          # counts 0, 5, 13, 29, 31 (golden_util.VARIANTS)
            for v in verts:
                cnt = oi.neighbor_count(v)
                s, m, e, lo, lo1 = ix.fastscan_block(lut, qp, v, dqp)
                os_, om, oe, olo, olo1 = oi.fastscan_vertex(lut, qp, v, dqp)
                assert np.array_equal(s[:cnt], os_[:cnt]) and np.array_equal(m[:cnt], om[:cnt]), (name, bits, v)
                assert _beq(e[:cnt], oe[:cnt]) and _beq(lo[:cnt], olo[:cnt]) and _beq(lo1[:cnt], olo1[:cnt]), (name, bits, v, dqp, cnt)
                if cnt < 32:
                    continue
                if bits > 1:
                    # stage-2 skip: result heap full and a threshold below every stage-1 bound
                    thr = float(np.min(olo1)) * 0.5
                    if thr > 0:
                        _, _, e2, lo2, _ = ix.fastscan_block(lut, qp, v, dqp, worst=thr, nn_full=True)
                        assert np.all(e2 == np.float32(3.402823466e+38)) and _beq(lo2, olo1)
                    thr = float(np.min(olo1)) * 1.5 + 1e-3
                    _, _, e3, lo3, _ = ix.fastscan_block(lut, qp, v, dqp, worst=thr, nn_full=True)
                    assert _beq(e3, oe) and _beq(lo3, olo)
        # many random parent distances: sqrt and the divisions must be correctly rounded
        qp = np.array([co[0], co[1], co[2], 0.97, 0.01, 0.2, 0.07], np.float32)
        for dqp in rng.uniform(0.01, 3e4, 40).astype(np.float32):
            v = int(rng.integers(0, oi.n))
            cnt = oi.neighbor_count(v)
            s, m, e, lo, lo1 = ix.fastscan_block(lut, qp, v, float(dqp))
            os_, om, oe, olo, olo1 = oi.fastscan_vertex(lut, qp, v, float(dqp))
            assert _beq(e[:cnt], oe[:cnt]) and _beq(lo[:cnt], olo[:cnt]) and _beq(lo1[:cnt], olo1[:cnt]), (name, bits, v, dqp)
        ids = rng.integers(0, oi.n, 50).astype(np.uint32)
        assert _beq(ix.exact_l2(Q[qi], ids), oi.exact_l2(Q[qi], ids))


@pytest.mark.parametrize("D,bits", [(16, 1), (16, 4), (64, 2), (128, 1), (128, 2), (128, 4),
                                    (256, 1), (512, 2), (1024, 1), (1024, 2), (1024, 4), (2048, 1),
                                    (2048, 4)])
def test_stream_blocks_match_oracle(cph, oracle, D, bits):
    st = cph.FastScanStream(D, bits, 300, seed=9)
    L = oracle.layout(D, bits)
    nb_bytes = L[0] - L[1]
    blocks, lut, qp, dqp = st.export(0, 300, nb_bytes)
    est, lower = st.eval(0, 300)
    blocks = blocks.reshape(300, nb_bytes)
    for b in (0, 1, 17, 299):
        nb = blocks[b]
        planes = nb[L[2]:L[2] + bits * D * 4].reshape(bits, D // 8, 32)
        nop = nb[L[3]:L[3] + 128].view(np.float32)
        ipqo = nb[L[4]:L[4] + 128].view(np.float32)
        ipcp = nb[L[5]:L[5] + 128].view(np.float32)
        pop = nb[L[6]:L[6] + 64].view(np.uint16)
        if bits == 1:
            s = oracle.fastscan_plane(D, lut, planes[0])
            e, lo = oracle.convert_1bit(D, qp, s, nop, ipqo, ipcp, pop, dqp)
        else:
            wpop = nb[L[7]:L[7] + 64].view(np.uint16)
            s, m = oracle.fastscan_nbit(D, bits, lut, planes)
            e, lo = oracle.convert_nbit(D, bits, qp, s, m, nop, ipqo, ipcp, pop, wpop, dqp)
        assert _beq(est[b], e) and _beq(lower[b], lo), (D, bits, b)
    ms, ck = st.run(2)
    assert ms > 0 and np.isfinite(ck)
    st.close()
    if D == 128 and bits <= 2:
        # the two-blocks-per-wave kernel of the narrow shapes: an odd block count and an odd first block
        st = cph.FastScanStream(D, bits, 301, seed=11)
        blocks, lut, qp, dqp = st.export(297, 4, nb_bytes)
        est, lower = st.eval(297, 4)
        for b in range(4):
            nb = blocks.reshape(4, nb_bytes)[b]
            planes = nb[L[2]:L[2] + bits * D * 4].reshape(bits, D // 8, 32)
            nop, ipqo, ipcp = (nb[L[j]:L[j] + 128].view(np.float32) for j in (3, 4, 5))
            pop = nb[L[6]:L[6] + 64].view(np.uint16)
            if bits == 1:
                e, lo = oracle.convert_1bit(D, qp, oracle.fastscan_plane(D, lut, planes[0]), nop, ipqo, ipcp, pop, dqp)
            else:
                s_, m_ = oracle.fastscan_nbit(D, bits, lut, planes)
                e, lo = oracle.convert_nbit(D, bits, qp, s_, m_, nop, ipqo, ipcp, pop, nb[L[7]:L[7] + 64].view(np.uint16), dqp)
            assert _beq(est[b], e) and _beq(lower[b], lo), (D, bits, b)
        st.close()


@pytest.mark.parametrize("name,bits,D,dim", [("g128", 1, 128, 128), ("sift96", 4, 128, 96),
                                             ("g16", 2, 16, 10), ("g1024", 2, 1024, 960)])
def test_device_query_encoder_matches_reference(cph, gold, name, bits, D, dim):
    """The on-device rotation + LUT scalars + coefficients against the reference's vectors."""
    ix = _load(cph, name, bits)
    q = gold[f"E/{D}/{dim}/q"]
    for i in range(len(q)):
        lut, co = ix.encode_query(q[i])
        assert np.array_equal(lut, gold[f"E/{D}/{dim}/lut"][i]), (name, i)
        assert _beq(co, gold[f"E/{D}/{dim}/coeffs"][i]), (name, i, co, gold[f"E/{D}/{dim}/coeffs"][i])


def test_save_roundtrip_is_byte_identical(cph, tmp_path):
    ix = _load(cph, "g128", 4)
    p = tmp_path / "out.idx"
    ix.save(str(p))
    assert p.read_bytes() == open(fixture_path("g128", 4), "rb").read()
    assert ix.size == DATASETS["g128"]["n"] and ix.dim == 128 and ix.is_finalized


def test_api_errors(cph, tmp_path):
    with pytest.raises(ValueError, match="Unsupported bits=3"):
        cph.CPIndex(128, 3)
    with pytest.raises(ValueError, match="Unsupported dimension 4096"):
        cph.CPIndex(4096, 1)
    ix = cph.CPIndex(128, 4)
    assert not ix.is_finalized and ix.size == 0
    with pytest.raises(RuntimeError, match="must be finalized"):
        ix.save(str(tmp_path / "x.idx"))
    with pytest.raises(RuntimeError):
        ix.search(np.zeros(128, np.float32), 10)
    with pytest.raises(ValueError, match="query must be 1D"):
        ix.search(np.zeros((2, 128), np.float32), 10)
    with pytest.raises(ValueError, match="queries must be"):
        ix.search_batch(np.zeros((2, 64), np.float32), 10)
    bad = tmp_path / "bad.idx"
    bad.write_bytes(b"\0" * 200)
    with pytest.raises(RuntimeError, match="Invalid magic"):
        ix.load(str(bad))
    with pytest.raises(RuntimeError, match="template parameters mismatch"):
        ix.load(fixture_path("g128", 2))
    trunc = tmp_path / "trunc.idx"
    trunc.write_bytes(open(fixture_path("g128", 4), "rb").read()[:100000])
    with pytest.raises(RuntimeError, match="truncated"):
        ix.load(str(trunc))
    assert not ix.is_finalized  # failed loads leave the index untouched
    ix.load(fixture_path("g128", 4))
    ids, d = ix.search_batch(np.zeros((0, 128), np.float32), 10)
    assert ids.shape == (0, 10)
    ids, d = ix.search(np.ones(128, np.float32), 0)   # k clamped to >= 1
    assert len(ids) == 1


def test_capacity_overflow_rerun_is_exact(cph, gold):
    """A tiny per-slot capacity forces the overflow -> full-capacity re-run path."""
    ix = _load(cph, "g128", 2)
    ix.set_search_params(slots=8, beam_capacity=64)
    Q = gold["Q/g128"]
    ids, d = ix.search_batch(Q, 10)
    assert ix.last_search_stats()["rerun_queries"] > 0
    assert np.array_equal(ids, gold["S/g128/b2/plain/k10/ids"]) and _beq(d, gold["S/g128/b2/plain/k10/d"])


@pytest.mark.parametrize("name,bits", [("g128", 4), ("sift96", 4), ("g16", 1)])
def test_launch_order_and_per_query_work(cph, oracle, gold, name, bits):
    """More queries than resident slots: the closest-entry-first launch order changes nothing in
    the results, and the per-query expansion counts equal the oracle's counters."""
    ix = _load(cph, name, bits)
    ix.set_search_params(slots=4, beam_capacity=0)
    Q = gold[f"Q/{name}"]
    ids, d = ix.search_batch(Q, 10)
    assert np.array_equal(ids, gold[f"S/{name}/b{bits}/plain/k10/ids"])
    assert _beq(d, gold[f"S/{name}/b{bits}/plain/k10/d"])
    st = ix.last_search_stats()
    assert st["rerun_queries"] == 0 and st["slots"] == 4
    work = ix.last_query_expansions(len(Q))
    assert int(work.sum()) == st["expansions"]
    oi = oracle.load(fixture_path(name, bits))
    _, _, _, ctr = oi.search_batch(Q, 10, counters=True)
    assert np.array_equal(work.astype(np.uint64), ctr[:, 0])
    with pytest.raises(ValueError):
        ix.last_query_expansions(len(Q) + 1)


def test_large_index_against_oracle(cph, oracle, tmp_path):
    """Bigger graph (many thousands of expansions per query): GPU vs the oracle on an index built
    by the compiled reference when it is available on this box."""
    from oracle_lib import ref_available, ref_module
    if not ref_available():
        pytest.skip("oracle/_ref not present on this box")
    m = ref_module()
    rng = np.random.default_rng(77)
    n, dim = 20000, 128
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((64, dim)).astype(np.float32)
    for bits in (1, 4):
        ridx = m.CPIndex(dim, bits)
        ridx.build(X)
        ridx.finalize()
        p = str(tmp_path / f"big_{bits}.idx")
        ridx.save(p)
        ix = cph.CPIndex(dim, bits)
        ix.load(p)
        for k in (10, 100):
            rids, rd = ridx.search_batch(Q, k)
            ids, d = ix.search_batch(Q, k)
            assert np.array_equal(ids, rids), (bits, k)
            assert _beq(d, rd), (bits, k)


@pytest.mark.parametrize("bits,patch", [(4, None), (4, dict(search_gamma=1.15, gamma_max=1.6, gamma_beta=0.6, gamma_warmup=5)),
                                        (2, dict(search_gamma=1.15, gamma_max=1.6, gamma_beta=0.6, gamma_warmup=5)), (1, None)])
def test_search_trajectory_counters_at_scale(cph, oracle, tmp_path, bits, patch):
    """Not only the top-k: the search PATH.  Per-query expansion counts and the batch totals of new neighbours, beam pushes
    and skipped stage-2 batches against the oracle's counters on a 60,000-vertex 4-bit index, 1,500 queries (~2 M
    expansions; 20,000 vertices and 600 queries for the narrow codes, whose searches run longer) -- for the probe-first
    instantiation (the first batch on a D = 128 index: it sees only the NEW neighbours' codes and has to reproduce the
    reference's stage-2 decision over ALL of a list, rabitq_search.hpp:178-187) and for the instantiation without it (the
    small-batch launch).  With the builder's calibration (gamma ~ 1e9: DABS off) and with a
    finite gamma patched in (DABS and gamma-termination on)."""
    from golden_util import apply_patch
    rng = np.random.default_rng(4242)
    n, dim, k = (60000 if bits == 4 else 20000), 128, 10
    nq = 1500 if bits == 4 else 600
    X = rng.standard_normal((n, dim)).astype(np.float32)        # unclustered: ~1,000 expansions per query at 4 bits
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    ix = cph.CPIndex(dim, bits)
    ix.build(X)
    ix.finalize()
    p = str(tmp_path / "traj.idx")
    ix.save(p)
    if patch:
        data = open(p, "rb").read()
        open(p, "wb").write(apply_patch(data, patch))
        ix = cph.CPIndex(dim, bits)
        ix.load(p)
    oi = oracle.load(p)
    oids, od, _, ctr = oi.search_batch(Q, k, nthreads=16, counters=True)
    # -- the batch path: probe first
    ids, d = ix.search_batch(Q, k)
    st = ix.last_search_stats()
    work = ix.last_query_expansions(len(Q))
    assert np.array_equal(ids, oids) and _beq(d, od)
    assert np.array_equal(work.astype(np.uint64), ctr[:, 0]), int((work != ctr[:, 0]).sum())
    assert st["expansions"] == int(ctr[:, 0].sum())
    assert st["new_neighbours"] == int(ctr[:, 3].sum())
    assert st["beam_pushes"] == int(ctr[:, 4].sum()) - len(Q)          # (the oracle counts the entry's push)
    assert st["exact_l2"] >= int(ctr[:, 1].sum())                       # speculative reranks: a superset
    skipped = int(ctr[:, 6].sum())
    assert st["stage2_skipped"] <= skipped <= st["stage2_skipped"] + st["stage2_undecided"], (st, skipped)
    print("trajectory:", {"bits": bits, "patch": bool(patch), **st, "oracle_stage2_skipped": skipped})
    # -- the small-batch launch: the instantiation without probe first decides every batch
    tot = dict(expansions=0, new_neighbours=0, beam_pushes=0, stage2_skipped=0, stage2_undecided=0, stage2_reruns=0)
    for lo in range(0, 320, 32):
        ids2, d2 = ix.search_batch(Q[lo:lo + 32], k)
        st2 = ix.last_search_stats()
        assert np.array_equal(ids2, oids[lo:lo + 32]) and _beq(d2, od[lo:lo + 32])
        assert np.array_equal(ix.last_query_expansions(32).astype(np.uint64), ctr[lo:lo + 32, 0])
        for key in tot:
            tot[key] += st2[key]
    assert tot["expansions"] == int(ctr[:320, 0].sum()) and tot["new_neighbours"] == int(ctr[:320, 3].sum())
    assert tot["beam_pushes"] == int(ctr[:320, 4].sum()) - 320
    assert tot["stage2_skipped"] == int(ctr[:320, 6].sum()) and tot["stage2_undecided"] == 0 and tot["stage2_reruns"] == 0


@pytest.mark.parametrize("n_fill", [1, 2, 3, 7, 254, 255, 256, 257, 300, 511, 512, 4095, 4096, 4097, 8191, 8192, 9000, 16384, 70000, 262143, 262144, 270000])
def test_beam_heap_routines_move_like_libstdcxx(cph, oracle, n_fill):
    """The beam's wave-parallel heap routines (heap_push_wave / heap_pop_wave in LDS, beam_push_hybrid / beam_pop_hybrid
    once it spills to HBM: windows of five levels) against libstdc++'s std::push_heap / std::pop_heap on the same
    operation list: n_fill pushes, then a random mix of pops and pushes, then pops down across the LDS boundary.  Keys
    are drawn from few distinct values, so equal keys -- whose order a heap's exact element movement decides -- are
    everywhere.  The whole heap array must match after the last operation, element for element."""
    rng = np.random.default_rng(1000 + n_fill)
    for distinct in (8, 1 << 20):
        # 0 = pop, 1 = push, 2 = pop-then-push the way one expansion does it (the leaf's ancestors fetched ahead of the pop
        # into LDS, the push reading them from there unless the pop's path crossed them)
        mix = rng.integers(0, 3, size=900).astype(np.uint8)
        drain = np.zeros(min(n_fill + 200, 700), np.uint8)
        refill = np.ones(300, np.uint8)
        tail = rng.choice(np.array([0, 1, 1, 2, 2, 2], np.uint8), size=600)
        ops = np.concatenate([np.ones(n_fill, np.uint8), mix, drain, refill, tail])
        n_push = int((ops != 0).sum())
        keys = rng.integers(0, distinct, size=n_push).astype(np.float32)
        ids = np.arange(n_push, dtype=np.uint32)
        gk, gi = cph.heap_ops_debug(ops, keys, ids)
        wk, wi = oracle.std_heap_ops(ops, keys, ids)
        assert len(gk) == len(wk), (n_fill, distinct, len(gk), len(wk))
        assert np.array_equal(gi, wi), (n_fill, distinct, int((gi != wi).sum()))
        assert gk.tobytes() == wk.tobytes()


@pytest.mark.parametrize("k", [256, 257, 300, 1000])
def test_large_k_result_heap(cph, gold, k):
    """k past the 256 entries the wave-parallel result-heap routines cover (kWaveHeapMax): the heap falls back to the
    lane-0 sift; k = 1000 also exceeds what the small fixture can return (padding with -1 / FLT_MAX)."""
    from oracle_lib import Oracle
    ix = _load(cph, "g128", 4)
    Q = gold["Q/g128"][:16]
    oids, od, _ = Oracle().load(fixture_path("g128", 4)).search_batch(Q, k)
    ids, d = ix.search_batch(Q, k)
    assert np.array_equal(ids, oids) and _beq(d, od)


def test_beams_beyond_the_lds_levels_match_oracle(cph, oracle, tmp_path):
    """Gaussian data at low bit width: thousands of expansions per query, beams of thousands of entries -- the
    heap levels past the 255 LDS entries live in HBM and are popped / pushed by the window routines
    (beam_pop_hybrid, beam_push_hybrid).  Index built by this repo's builder; ids and distance bits against the oracle
    (and the compiled reference when it is on the box), in throughput mode and in latency mode (batch <= slots)."""
    from oracle_lib import Oracle, ref_available, ref_module
    rng = np.random.default_rng(2024)
    n, dim, bits, k = 30000, 64, 2, 20
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((96, dim)).astype(np.float32)
    ix = cph.CPIndex(dim, bits)
    ix.build(X)
    ix.finalize()
    p = str(tmp_path / "spill.idx")
    ix.save(p)
    oids, od, _ = Oracle().load(p).search_batch(Q, k)
    ids, d = ix.search_batch(Q, k)
    st = ix.last_search_stats()
    assert st["expansions"] / len(Q) > 1500, st          # the regime the test is about
    assert st["beam_pushes"] / len(Q) > 2000, st
    assert np.array_equal(ids, oids) and _beq(d, od)
    ix.set_search_params(slots=64, beam_capacity=0)       # 96 queries on 64 slots: a queue, no latency mode
    ids2, d2 = ix.search_batch(Q, k)
    assert np.array_equal(ids2, oids) and _beq(d2, od)
    if ref_available():
        r = ref_module().CPIndex(dim, bits)
        r.load(p)
        rids, rd = r.search_batch(Q, k)
        assert np.array_equal(ids, rids) and _beq(d, rd)


@pytest.mark.parametrize("name,bits", [("g128", 4), ("g16", 2)])
def test_repeated_neighbour_ids(cph, oracle, gold, tmp_path, name, bits):
    """A graph whose neighbour lists repeat an id (the reference never writes one, the loader flags
    it): only the first copy of the id is new.  Fixture patched in place, GPU against the oracle."""
    import gzip
    from golden_util import vertex_layout
    dim = DATASETS[name]["dim"]
    n = DATASETS[name]["n"]
    D = max(16, 1 << (dim - 1).bit_length())
    src = fixture_path(name, bits)
    data = bytearray(gzip.open(src, "rb").read() if src.endswith(".gz") else open(src, "rb").read())
    vb, nb_off, codes = vertex_layout(D, bits)
    base = 68 + 248 + 72 + dim * 4 + n * 4 + n * 4 + n * D * 4
    ids_off = nb_off + codes + 3 * 128 + 64 + (64 if bits > 1 else 0)
    patched = 0
    for v in range(0, n, 3):
        o = base + v * vb + ids_off
        cnt = int.from_bytes(data[o + 128:o + 132], "little")
        if cnt >= 8:
            data[o + 5 * 4:o + 6 * 4] = data[o + 2 * 4:o + 3 * 4]      # slot 5 repeats slot 2
            data[o + 7 * 4:o + 8 * 4] = data[o + 0 * 4:o + 1 * 4]      # slot 7 repeats slot 0
            patched += 1
    assert patched > 50
    p = str(tmp_path / "dups.idx")
    open(p, "wb").write(bytes(data))
    oi = oracle.load(p)
    ix = cph.CPIndex(dim, bits)
    ix.load(p)
    Q = gold[f"Q/{name}"]
    for k in (10, 100):
        oids, od, ocnt = oi.search_batch(Q, k)
        ids, d = ix.search_batch(Q, k)
        assert np.array_equal(ids, oids), (name, bits, k)
        assert _beq(d, od), (name, bits, k)


@pytest.mark.parametrize("n", [1, 63, 1024, 10000, 70001])
def test_order_kernel_is_a_sorted_permutation(cph, n):
    """The counting sort behind the launch order: a permutation, ascending in the 14-bit bucket of
    the key; zeros, denormals, huge values, infinities, negative and NaN keys included."""
    ix = cph.CPIndex(16, 1)
    rng = np.random.default_rng(n)
    keys = (rng.gamma(2.0, 5000.0, n)).astype(np.float32)
    special = np.array([0.0, 1e-45, 1e-30, 3.4e38, np.inf, -1.0, -0.0, np.nan], np.float32)
    keys[: min(n, len(special))] = special[: min(n, len(special))]
    order = ix.order_queries(keys)
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
    bits = keys.view(np.uint32)
    bucket = np.where(bits >> 31, 0, np.minimum(bits >> 17, 16383)).astype(np.int64)
    b = bucket[order]
    assert (b[1:] >= b[:-1]).all()


def test_results_do_not_depend_on_launch_order(gold):
    """CPH_QUERY_ORDER=0 (queries handed out in batch order) against the same goldens, with fewer
    slots than queries so that the order matters for the schedule."""
    import subprocess
    import sys
    import os
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
from golden_util import DATASETS, fixture_path, golden
import cphnsw_mi355x
g = golden()
for name, bits in (("g128", 4), ("sift96", 4), ("g16", 2)):
    ix = cphnsw_mi355x.CPIndex(DATASETS[name]["dim"], bits)
    ix.load(fixture_path(name, bits))
    ix.set_search_params(slots=4, beam_capacity=0)
    ids, d = ix.search_batch(g[f"Q/{name}"], 10)
    assert np.array_equal(ids, g[f"S/{name}/b{bits}/plain/k10/ids"]), (name, bits)
    assert d.tobytes() == g[f"S/{name}/b{bits}/plain/k10/d"].tobytes(), (name, bits)
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CPH_QUERY_ORDER="0")
    out = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + code], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_device_batches_on_two_streams_overlap_and_match(cph, gold):
    """search_batch_device never waits for the device: batches enqueued alternately on two streams
    (each call gets the other scratch set) give the goldens, also when a tiny capacity sends
    queries through the on-device full-capacity re-run launch."""
    import torch
    dev = torch.device("cuda", 0)
    for cap in (0, 64):
        ix = _load(cph, "g128", 4)
        ix.set_search_params(slots=8 if cap else 0, beam_capacity=cap)
        Q = torch.from_numpy(gold["Q/g128"]).to(dev)
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        torch.cuda.synchronize()
        outs = []
        for i in range(6):
            st = streams[i & 1]
            st.wait_stream(torch.cuda.current_stream(dev))
            outs.append(ix.search_batch_device(Q, 10 if i % 3 else 100, stream=st))
        ix.synchronize()
        for i, (ids, d) in enumerate(outs):
            k = 10 if i % 3 else 100
            assert np.array_equal(ids.cpu().numpy(), gold[f"S/g128/b4/plain/k{k}/ids"]), (cap, i)
            assert _beq(d.cpu().numpy(), gold[f"S/g128/b4/plain/k{k}/d"]), (cap, i)
        st = ix.last_search_stats()
        assert (st["rerun_queries"] > 0) == (cap == 64) and st["capacity"] == (64 if cap else DATASETS["g128"]["n"] + 1)


def test_device_batch_validates_out_buffers(cph, gold):
    import torch
    dev = torch.device("cuda", 0)
    ix = _load(cph, "g128", 4)
    Q = torch.from_numpy(gold["Q/g128"]).to(dev)
    n = Q.shape[0]
    good = (torch.empty((n, 10), dtype=torch.int64, device=dev), torch.empty((n, 10), dtype=torch.float32, device=dev))
    ids, d = ix.search_batch_device(Q, 10, out=good)
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy(), gold["S/g128/b4/plain/k10/ids"])
    for bad in ((good[0][:, :5], good[1]), (good[0].to(torch.int32), good[1]), (good[0].cpu(), good[1]),
                (good[0], torch.empty((n, 20), dtype=torch.float32, device=dev)[:, ::2])):
        with pytest.raises(ValueError, match="out must be"):
            ix.search_batch_device(Q, 10, out=bad)
    with pytest.raises(ValueError):
        ix.search_batch_device(Q.cpu(), 10)


@pytest.mark.parametrize("name,bits", [("g128", 4), ("g16", 2), ("g1024", 2)])
def test_native_file_roundtrip(cph, gold, tmp_path, name, bits):
    """GPU-native index file: v2 -> native -> load_native gives the same search results and stored vectors,
    and a v2 file written from it is searched identically (by us; and it has the v2 file's size)."""
    ix = _load(cph, name, bits)
    pn = str(tmp_path / "x.cphn")
    ix.save_native(pn)
    ix2 = cph.CPIndex(DATASETS[name]["dim"], bits)
    ix2.load_native(pn)
    assert ix2.is_finalized and ix2.size == DATASETS[name]["n"]
    Q = gold[f"Q/{name}"]
    for k in (10, 100):
        ids, d = ix2.search_batch(Q, k)
        assert np.array_equal(ids, gold[f"S/{name}/b{bits}/plain/k{k}/ids"]) and _beq(d, gold[f"S/{name}/b{bits}/plain/k{k}/d"])
    assert _beq(ix2.get_vectors(), ix.get_vectors())
    p2 = tmp_path / "back.idx"
    ix2.save(str(p2))
    assert len(p2.read_bytes()) == len(open(fixture_path(name, bits), "rb").read())
    ix3 = cph.CPIndex(DATASETS[name]["dim"], bits)
    ix3.load(str(p2))
    ids3, d3 = ix3.search_batch(Q, 10)
    assert np.array_equal(ids3, gold[f"S/{name}/b{bits}/plain/k10/ids"]) and _beq(d3, gold[f"S/{name}/b{bits}/plain/k10/d"])
    with pytest.raises(RuntimeError, match="not a CP-HNSW MI355X native"):
        ix3.load_native(fixture_path(name, bits))
    with pytest.raises(RuntimeError, match="mismatch"):
        cph.CPIndex(DATASETS[name]["dim"], 1 if bits != 1 else 2).load_native(pn)


def test_native_file_save_over_its_own_mapping_and_rejects_corruption(cph, gold, tmp_path):
    """load_native(p) serves vectors and own-code headers out of a mapping of p: save_native(p) and save(p) on the same
    path must neither crash (a truncating open would SIGBUS the mapping) nor lose the file -- both writers go through a
    temporary file and rename().  A truncated or corrupted native file is an error code, not a fault: the header's
    offsets, the neighbour counts and the neighbour ids are all checked before anything reaches the GPU."""
    name, bits = "g128", 4
    ix = _load(cph, name, bits)
    pn = str(tmp_path / "x.cphn")
    ix.save_native(pn)
    good = open(pn, "rb").read()
    a = cph.CPIndex(DATASETS[name]["dim"], bits)
    a.load_native(pn)
    a.save_native(pn)                                  # over the file the handle is mapped from
    assert open(pn, "rb").read() == good
    Q = gold[f"Q/{name}"]
    ids, d = a.search_batch(Q, 10)
    assert np.array_equal(ids, gold[f"S/{name}/b{bits}/plain/k10/ids"]) and _beq(d, gold[f"S/{name}/b{bits}/plain/k10/d"])
    assert _beq(a.get_vectors(), ix.get_vectors())     # still served from the (old) mapping
    a.save(pn)                                         # a v2 file under the same name
    assert open(pn, "rb").read() == open(fixture_path(name, bits), "rb").read()
    assert _beq(a.get_vectors(), ix.get_vectors())
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    # malformed native files: the handle stays searchable with what it had
    import struct
    n = DATASETS[name]["n"]
    hdr_n, hdr_blocks = 24, 104                        # offsetof(NativeHeader, n / blocks_off)
    blocks_off = struct.unpack_from("<Q", good, hdr_blocks)[0]
    stride = struct.unpack_from("<I", good, 32)[0]
    cases = {
        "truncated": good[: len(good) // 2],
        "header only": good[:120],
        "n too large": good[:hdr_n] + struct.pack("<Q", n * 50) + good[hdr_n + 8:],
        "blocks_off beyond the file": good[:hdr_blocks] + struct.pack("<Q", len(good) + 4096) + good[hdr_blocks + 8:],
        "neighbour id out of range": good[:blocks_off + 7 * stride + 2048 + 512] + struct.pack("<I", n + 9)
                                     + good[blocks_off + 7 * stride + 2048 + 512 + 4:],
        "neighbour count 77": good[:blocks_off + 7 * stride + 2048 + 512 + 128] + struct.pack("<I", 77)
                              + good[blocks_off + 7 * stride + 2048 + 512 + 128 + 4:],
    }
    for what, blob in cases.items():
        pb = tmp_path / "bad.cphn"
        pb.write_bytes(blob)
        with pytest.raises(RuntimeError):
            a.load_native(str(pb))
        ids2, d2 = a.search_batch(Q, 10)
        assert np.array_equal(ids2, ids) and _beq(d2, d), what


def test_torch_can_initialise_after_the_library():
    """Process-level: our library first, torch.cuda afterwards (a PyTorch-ROCm wheel bundles its own HIP runtime;
    cphnsw_mi355x._lib makes both use the same one)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import cphnsw_mi355x as c; s = c.FastScanStream(128, 4, 64); s.run(1); s.close(); "
            "import torch; torch.cuda.init(); assert torch.cuda.device_count() >= 1; "
            "x = torch.ones(8, device='cuda').sum().item(); assert x == 8.0; print('OK')") % os.path.join(root, "rabitq-ann-search_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def _threaded_search(ix, Q, k, n_threads, calls):
    """n_threads Python threads, each calling ix.search() `calls` times (ctypes releases the GIL around the C call);
    returns {(thread, call): (ids, dist)} and the wall time."""
    import threading
    import time
    out, errs = {}, []
    start = threading.Barrier(n_threads + 1)

    def work(t):
        try:
            start.wait()
            for c in range(calls):
                qi = (t * calls + c) % len(Q)
                out[(t, c)] = (qi,) + ix.search(Q[qi], k)
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    start.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    el = time.perf_counter() - t0
    assert not errs, errs[:3]
    return out, el


def test_concurrent_search_calls_are_coalesced_and_exact(cph, gold, tmp_path):
    """The reference's search() takes a SHARED lock and releases the GIL (src/bindings.cpp:146-175,
    api/hnsw_index.hpp:172): T threads search one index in parallel.  Here concurrent callers are gathered into one
    launch by whoever finds none in flight (cph_search: leader / followers).  16 threads x 200 calls on a fixture and
    on a 70,000-vertex index: every answer equals the batch path's row (unpadded), mixed k included."""
    ix = _load(cph, "g128", 4)
    Q = gold["Q/g128"]
    ref = {k: ix.search_batch(Q, k) for k in (1, 10, 20)}
    for k in (10, 1):
        out, _ = _threaded_search(ix, Q, k, 16, 200)
        assert len(out) == 16 * 200
        for (_, _), (qi, ids, d) in out.items():
            m = int((ref[k][0][qi] >= 0).sum())
            assert np.array_equal(ids, ref[k][0][qi, :m]) and _beq(d, ref[k][1][qi, :m])
    # mixed k: callers with different k never share a launch
    import threading
    res, errs = {}, []

    def work(t):
        try:
            k = (1, 10, 20)[t % 3]
            for c in range(60):
                qi = (7 * t + c) % len(Q)
                res[(t, c)] = (k, qi) + ix.search(Q[qi], k)
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs[:3]
    for (k, qi, ids, d) in res.values():
        m = int((ref[k][0][qi] >= 0).sum())
        assert np.array_equal(ids, ref[k][0][qi, :m]) and _beq(d, ref[k][1][qi, :m])
    # an index of some size: the launches are long enough for callers to pile up behind them
    rng = np.random.default_rng(99)
    n, dim = 70000, 128
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Qb = rng.standard_normal((400, dim)).astype(np.float32)
    big = cph.CPIndex(dim, 4)
    big.build(X)
    big.finalize()
    rids, rd = big.search_batch(Qb, 10)
    out, el = _threaded_search(big, Qb, 10, 16, 200)
    for (_, _), (qi, ids, d) in out.items():
        m = int((rids[qi] >= 0).sum())
        assert np.array_equal(ids, rids[qi, :m]) and _beq(d, rd[qi, :m])
    print(f"concurrent search: 16 threads x 200 calls on 70k x 128 / 4-bit: {16 * 200 / el:.0f} QPS")
    # errors reach every caller of a group: an unfinalized index
    empty = cph.CPIndex(128, 4)
    with pytest.raises(RuntimeError):
        empty.search(np.zeros(128, np.float32), 10)


@pytest.mark.parametrize("bits", [4, 1])
def test_probe_first_at_d1024_matches_oracle(cph, oracle, tmp_path, bits):
    """The D = 1024 instantiations with probe first (GIST1M-class shape; the committed D = 1024 fixture is 2-bit): a
    4,000-vertex index of 960-dim clustered rows built here, the batch path (probe first + re-run launch) and the
    small-batch launch (whole blocks) against the oracle -- ids, distance bits, per-query expansion counts, totals."""
    rng = np.random.default_rng(960 + bits)
    n, dim, k = 4000, 960, 10
    cent = rng.random((40, dim)).astype(np.float32)
    X = (cent[rng.integers(0, 40, n)] + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
    Q = (cent[rng.integers(0, 40, 200)] + 0.05 * rng.standard_normal((200, dim))).astype(np.float32)
    ix = cph.CPIndex(dim, bits)
    ix.build(X)
    ix.finalize()
    p = str(tmp_path / "d1024.idx")
    ix.save(p)
    oi = oracle.load(p)
    oids, od, _, ctr = oi.search_batch(Q, k, nthreads=8, counters=True)
    ids, d = ix.search_batch(Q, k)                        # 200 queries: the general path
    st = ix.last_search_stats()
    assert np.array_equal(ids, oids) and _beq(d, od)
    assert np.array_equal(ix.last_query_expansions(len(Q)).astype(np.uint64), ctr[:, 0])
    assert st["new_neighbours"] == int(ctr[:, 3].sum()) and st["beam_pushes"] == int(ctr[:, 4].sum()) - len(Q)
    ids2, d2 = ix.search_batch(Q[:24], k)                 # the small-batch launch
    assert np.array_equal(ids2, oids[:24]) and _beq(d2, od[:24])
    assert np.array_equal(ix.last_query_expansions(24).astype(np.uint64), ctr[:24, 0])
