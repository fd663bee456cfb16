"""Benchmark harness: loaders on CPU, the full protocol on the GPU."""
import json

import numpy as np
import pytest


def test_vecs_roundtrip(tmp_path):
    from cphnsw_mi355x.datasets import read_vecs, write_vecs
    rng = np.random.default_rng(0)
    a = rng.standard_normal((37, 24)).astype(np.float32)
    g = rng.integers(0, 1000, (37, 100)).astype(np.int32)
    write_vecs(tmp_path / "a.fvecs", a)
    write_vecs(tmp_path / "g.ivecs", g)
    assert np.array_equal(read_vecs(tmp_path / "a.fvecs", np.float32), a)
    assert np.array_equal(read_vecs(tmp_path / "g.ivecs", np.int32), g)


def test_recall_helpers():
    from cphnsw_mi355x.eval import dedup_first, recall_at_k
    ids = np.array([[3, 3, 5, 5, 7, -1], [1, 2, 3, 4, 5, 6]])
    assert dedup_first(ids, 3).tolist() == [[3, 5, 7], [1, 2, 3]]
    gt = np.array([[3, 5, 9], [9, 9, 9]])
    assert recall_at_k(dedup_first(ids, 3), gt, 3) == pytest.approx((2 / 3 + 0) / 2)


@pytest.mark.gpu
def test_run_benchmark_end_to_end(tmp_path):
    from cphnsw_mi355x.datasets import write_vecs
    from cphnsw_mi355x.eval import run_benchmark
    rng = np.random.default_rng(4)
    n, dim, nq = 4000, 128, 100
    cent = rng.gamma(2, 15, (8, dim))
    base = np.clip(np.round(cent[rng.integers(0, 8, n)] + rng.normal(0, 12, (n, dim))), 0, 218).astype(np.float32)
    q = np.clip(np.round(cent[rng.integers(0, 8, nq)] + rng.normal(0, 12, (nq, dim))), 0, 218).astype(np.float32)
    d = ((q[:, None, :].astype(np.float64) - base[None]) ** 2).sum(-1)
    gt = np.argsort(d, axis=1, kind="stable")[:, :100].astype(np.int32)
    root = tmp_path / "data" / "sift1m"
    root.mkdir(parents=True)
    write_vecs(root / "sift_base.fvecs", base)
    write_vecs(root / "sift_query.fvecs", q)
    write_vecs(root / "sift_groundtruth.ivecs", gt)
    out = run_benchmark("sift1m", tmp_path / "data", k=100, n_runs=2, output_dir=tmp_path / "res", bit_widths=(4,))
    r = out["results"][0]
    for key in ("build_time_s", "memory_mb", "recall_at_1", "recall_at_10", "recall_at_100", "adr", "qps",
                "median_latency_us"):
        assert key in r
    # ADR can fall below 1: duplicate result slots repeat a near neighbour against a farther GT rank (F2)
    assert r["qps"] > 0 and 0.5 < r["adr"] < 2.0
    assert 0.0 < r["recall_at_10_dedup"] <= 1.0   # (the raw score double-counts duplicate hits, F2)
    assert r["recall_at_100"] > 0.2
    saved = json.loads((tmp_path / "res" / "sift1m_results.json").read_text())
    assert saved["metadata"]["n_base"] == n and saved["metadata"]["k"] == 100


def _harness_fixture():
    import os
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "harness.npz"))
    return z, json.loads(bytes(z["metrics_json"]).decode()), json.loads(bytes(z["reference_run_json"]).decode())


def test_scores_match_the_reference_harness():
    """cphnsw_mi355x.eval.score_as_reference on the ids the reference returned for the fixture index against
    what the reference's own recall_at_k / ADR code computed for them (tests/golden/make_golden_harness.py)."""
    from cphnsw_mi355x.eval import score_as_reference
    z, metrics, ref_run = _harness_fixture()
    got = score_as_reference(z["ids"], z["base"], z["queries"], z["groundtruth"], int(z["k"]))
    assert got == metrics
    # schema of a real reference run: every key it writes exists in ours (tested on the GPU below)
    assert set(ref_run["metadata"]) == {"timestamp", "dataset", "n_base", "n_queries", "dim", "metric", "k", "n_runs"}


@pytest.mark.gpu
def test_harness_on_reference_built_index_reproduces_reference_numbers(tmp_path):
    """The fixture index was built and searched by the reference; the GPU drop-in loaded from the same file
    must return the same ids and distances, hence the reference harness' numbers to the last digit; and a
    full run_benchmark writes a superset of the keys a real reference run wrote."""
    import gzip
    import os
    import cphnsw_mi355x
    from cphnsw_mi355x.datasets import write_vecs
    from cphnsw_mi355x.eval import run_benchmark, score_as_reference
    from golden_util import GOLDEN_DIR
    z, metrics, ref_run = _harness_fixture()
    p = tmp_path / "h.idx"
    p.write_bytes(gzip.open(os.path.join(GOLDEN_DIR, "idx_harness_b4.idx.gz"), "rb").read())
    ix = cphnsw_mi355x.CPIndex(128, 4)
    ix.load(str(p))
    ids, d = ix.search_batch(z["queries"], int(z["k"]))
    assert np.array_equal(ids, z["ids"]) and d.tobytes() == z["dist"].tobytes()
    assert score_as_reference(ids, z["base"], z["queries"], z["groundtruth"], int(z["k"])) == metrics
    root = tmp_path / "data" / "sift1m"
    root.mkdir(parents=True)
    write_vecs(root / "sift_base.fvecs", z["base"])
    write_vecs(root / "sift_query.fvecs", z["queries"])
    write_vecs(root / "sift_groundtruth.ivecs", z["groundtruth"])
    out = run_benchmark("sift1m", tmp_path / "data", k=int(z["k"]), n_runs=2, output_dir=tmp_path / "res", bit_widths=(4,))
    assert set(ref_run["metadata"]) <= set(out["metadata"])
    assert set(ref_run["results"][0]) <= set(out["results"][0])
    ours, theirs = out["results"][0], [r for r in ref_run["results"] if r["algorithm"].endswith("4bit")][0]
    # same protocol on the same data: the uncorrected scores land where the reference's own run did
    assert abs(ours["as_reference"]["recall_at_100"] - theirs["recall_at_100"]) < 0.1
    assert 0.5 < ours["as_reference"]["adr"] / theirs["adr"] < 2.0
