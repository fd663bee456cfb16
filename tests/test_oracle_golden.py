"""The oracle (CPU restatement) against the committed reference golden vectors — CPU only.

Everything here is bit-exact: integer sums, fp32 epilogues, exact arithmetic, the query
encoder and the full search results (ids AND distances, duplicates and ties included).
"""
import numpy as np
import pytest

from golden_util import DATASETS, KS, SHORT_COUNTS, fixture_path


def _beq(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("D,dim", [(16, 10), (32, 32), (64, 50), (128, 128), (128, 96),
                                   (256, 200), (512, 512), (1024, 960), (2048, 1536)])
def test_query_encoder(oracle, gold, D, dim):
    q = gold[f"E/{D}/{dim}/q"]
    for i in range(len(q)):
        lut, co, rot = oracle.encode_query(q[i], D)
        assert np.array_equal(lut, gold[f"E/{D}/{dim}/lut"][i])
        assert _beq(co, gold[f"E/{D}/{dim}/coeffs"][i])
        assert _beq(rot, gold[f"E/{D}/{dim}/rot"][i])


@pytest.mark.parametrize("D", [16, 128, 1024])
@pytest.mark.parametrize("bits", [1, 2, 4])
def test_fastscan_block(oracle, gold, D, bits):
    k = f"F/{D}/b{bits}"
    lut, planes = gold[f"{k}/lut"], gold[f"{k}/planes"]
    nop, ipqo, ipcp = gold[f"{k}/nop"], gold[f"{k}/ipqo"], gold[f"{k}/ipcp"]
    pop, wpop = gold[f"{k}/pop"], gold[f"{k}/wpop"]
    for i in range(planes.shape[0]):
        if bits == 1:
            s = oracle.fastscan_plane(D, lut, planes[i, 0])
            m = m2 = s
        else:
            s, m = oracle.fastscan_nbit(D, bits, lut, planes[i])
            m2 = oracle.fastscan_msb(D, bits, lut, planes[i])
        assert np.array_equal(s, gold[f"{k}/sums"][i])
        assert np.array_equal(m, gold[f"{k}/msb"][i])
        assert np.array_equal(m2, gold[f"{k}/msb2"][i])
        for a, qp in enumerate(gold[f"{k}/qps"]):
            for c, dqp in enumerate(gold[f"{k}/dqps"]):
                if bits == 1:
                    e, lo = oracle.convert_1bit(D, qp, s, nop[i], ipqo[i], ipcp[i], pop[i], dqp)
                    lo1 = lo
                else:
                    lo1 = oracle.convert_msb(D, bits, qp, m2, nop[i], ipqo[i], ipcp[i], pop[i], dqp)
                    e, lo = oracle.convert_nbit(D, bits, qp, s, m, nop[i], ipqo[i], ipcp[i],
                                                pop[i], wpop[i], dqp)
                assert _beq(e, gold[f"{k}/est"][a, c, i])
                assert _beq(lo, gold[f"{k}/lower"][a, c, i])
                assert _beq(lo1, gold[f"{k}/lower1"][a, c, i])
                # short neighbour lists: the scalar tails for count % 8 != 0 (fastscan_kernel.hpp:174-193, :324-345)
                for cnt in SHORT_COUNTS:
                    if bits == 1:
                        e, lo = oracle.convert_1bit(D, qp, s, nop[i], ipqo[i], ipcp[i], pop[i], dqp, cnt)
                        lo1 = lo
                    else:
                        lo1 = oracle.convert_msb(D, bits, qp, m2, nop[i], ipqo[i], ipcp[i], pop[i], dqp, cnt)
                        e, lo = oracle.convert_nbit(D, bits, qp, s, m, nop[i], ipqo[i], ipcp[i],
                                                    pop[i], wpop[i], dqp, cnt)
                    for got, name in ((e, "est"), (lo, "lower"), (lo1, "lower1")):
                        assert _beq(got[:cnt], gold[f"{k}/c{cnt}/{name}"][a, c, i, :cnt]), (D, bits, cnt, name)


@pytest.mark.parametrize("D", [16, 128, 1024])
def test_exact_arithmetic(oracle, gold, D):
    a, b = gold[f"X/{D}/a"], gold[f"X/{D}/b"]
    for i in range(len(a)):
        assert oracle.dot(a[i], b[i]).tobytes() == gold[f"X/{D}/dot"][i].tobytes()
        assert oracle.l2(a[i], b[i]).tobytes() == gold[f"X/{D}/l2"][i].tobytes()


CASES = [(n, b, v) for n, s in DATASETS.items() for b in s["bits"] for v in s["variants"]]


@pytest.mark.parametrize("name,bits,variant", CASES)
def test_search_matches_reference(oracle, gold, name, bits, variant):
    ix = oracle.load(fixture_path(name, bits, variant))
    Q = gold[f"Q/{name}"]
    for k in KS:
        ids, d, cnt = ix.search_batch(Q, k, nthreads=2)
        assert np.array_equal(ids, gold[f"S/{name}/b{bits}/{variant}/k{k}/ids"]), (name, bits, variant, k)
        assert _beq(d, gold[f"S/{name}/b{bits}/{variant}/k{k}/d"])
    # unpadded single-query results (src/bindings.cpp:146-175)
    ids, d, cnt = ix.search_batch(Q[:4], 10, nthreads=1)
    for qi in range(4):
        gi = gold[f"S1/{name}/b{bits}/{variant}/q{qi}/ids"]
        assert cnt[qi] == len(gi)
        assert np.array_equal(ids[qi, :cnt[qi]], gi)


def test_layout_formula(oracle):
    # SURVEY.md §5.4 table (sizes from the compiled reference)
    expect = {(128, 1): 1280, (128, 2): 1856, (128, 4): 2880, (1024, 1): 4928,
              (1024, 2): 9216, (1024, 4): 17664}
    for (D, b), sz in expect.items():
        assert oracle.layout(D, b)[0] == sz


def test_load_errors(oracle, tmp_path):
    p = tmp_path / "bad.idx"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(RuntimeError, match="Invalid magic"):
        oracle.load(str(p))
    with pytest.raises(RuntimeError, match="Cannot open"):
        oracle.load(str(tmp_path / "missing.idx"))


ENC_SHAPES = ((10, 16), (50, 64), (96, 128), (128, 128), (300, 512), (960, 1024))


@pytest.mark.parametrize("dim,D", ENC_SHAPES)
@pytest.mark.parametrize("bits", [1, 2, 4])
def test_edge_encoder(oracle, gold_build, dim, D, bits):
    """Data-side encoder (per-edge RaBitQ / CAQ codes, nop, ip_qo, ip_cp, popcounts): the oracle's
    restatement against what the reference computed, code values and float bits."""
    k = f"ENC/{dim}/{D}/b{bits}"
    for c in range(len(gold_build[f"{k}/parent"])):
        v, a, s = oracle.encode_edges(gold_build[f"{k}/parent"][c], gold_build[f"{k}/nbrs"][c], D, bits)
        assert np.array_equal(v, gold_build[f"{k}/values"][c])
        assert _beq(a, gold_build[f"{k}/aux"][c])
        assert np.array_equal(s, gold_build[f"{k}/pops"][c])
