#!/usr/bin/env python3
"""Golden fixture for the benchmark harness — authoring container only.

    python tests/golden/make_golden_harness.py  ->  tests/golden/harness.npz + idx_harness_b4.idx.gz

Runs the REAL reference harness: a scratch package directory under /tmp holds the compiled reference
module (oracle/_ref/_core*.so, built by oracle/Makefile) next to the reference's own cphnsw/*.py, imported
from where they lie in /root/reference (nothing of it enters the repo).  On a seeded SIFT-like dataset
written as .fvecs/.ivecs it records
  * what `cphnsw.eval.run_benchmark` wrote for it (schema + the reference's own numbers; its build is not
    reproducible across thread counts, so these pin the schema and the protocol, not values),
  * a reference-built 4-bit index of the same base vectors (saved with the reference's `save`),
    the ids / distances the reference returns for the query set on it at k = 100, and the metrics the
    reference's `recall_at_k` / ADR formula give for those ids -- these values our harness must reproduce
    exactly on that index file.
Everything written is data (inputs, an index file, outputs)."""
import glob
import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
REF = "/root/reference"


def dataset(seed=2024, n=800, nq=40, dim=128, ncl=8):
    rng = np.random.default_rng(seed)
    cent = rng.gamma(2, 15, (ncl, dim))
    base = np.clip(np.round(cent[rng.integers(0, ncl, n)] + rng.normal(0, 12, (n, dim))), 0, 218).astype(np.float32)
    q = np.clip(np.round(cent[rng.integers(0, ncl, nq)] + rng.normal(0, 12, (nq, dim))), 0, 218).astype(np.float32)
    d = ((q[:, None, :].astype(np.float64) - base[None]) ** 2).sum(-1)
    gt = np.argsort(d, axis=1, kind="stable")[:, :100].astype(np.int32)
    return base, q, gt


def main():
    from cphnsw_mi355x.datasets import write_vecs
    scratch = tempfile.mkdtemp(prefix="refpkg_")
    pkg = os.path.join(scratch, "cphnsw")
    os.makedirs(pkg)
    for f in ("__init__.py", "eval.py", "datasets.py"):
        os.symlink(os.path.join(REF, "cphnsw", f), os.path.join(pkg, f))
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "_core*.so"))[0]
    os.symlink(so, os.path.join(pkg, os.path.basename(so)))
    sys.path.insert(0, scratch)
    import cphnsw                      # the reference package
    from cphnsw import eval as ref_eval
    from pathlib import Path

    base, q, gt = dataset()
    data = os.path.join(scratch, "data", "sift1m")
    os.makedirs(data)
    write_vecs(os.path.join(data, "sift_base.fvecs"), base)
    write_vecs(os.path.join(data, "sift_query.fvecs"), q)
    write_vecs(os.path.join(data, "sift_groundtruth.ivecs"), gt)
    res_dir = Path(scratch) / "res"
    res_dir.mkdir()
    k, n_runs = 100, 2
    out = ref_eval.run_benchmark("sift1m", Path(scratch) / "data", k, n_runs, res_dir)
    saved = json.loads((res_dir / "sift1m_results.json").read_text())

    idx = cphnsw.CPIndex(dim=128, bits=4)
    idx.build(base)
    idx.finalize()
    raw = os.path.join(scratch, "h.idx")
    idx.save(raw)
    with gzip.GzipFile(os.path.join(HERE, "idx_harness_b4.idx.gz"), "wb", compresslevel=9, mtime=0) as g:
        g.write(open(raw, "rb").read())
    ids, dist = idx.search_batch(q, k=k)
    ids = np.asarray(ids)
    gt64 = gt.astype(np.int64)
    adr_k = min(k, ref_eval.ADR_K, gt.shape[1])
    gt_d = np.sum((base[gt64[:, :adr_k]] - q[:, None, :]) ** 2, axis=2)
    res_d = np.sum((base[ids[:, :adr_k].astype(np.int64)] - q[:, None, :]) ** 2, axis=2)
    metrics = {
        "recall_at_1": round(ref_eval.recall_at_k(ids, gt64, 1), 4),
        "recall_at_10": round(ref_eval.recall_at_k(ids, gt64, min(k, 10)), 4),
        "recall_at_100": round(ref_eval.recall_at_k(ids, gt64, min(k, 100)), 4),
        "adr": round(float(np.mean(res_d / np.maximum(gt_d, ref_eval.ADR_EPS))), 6),
    }
    np.savez_compressed(os.path.join(HERE, "harness.npz"), base=base, queries=q, groundtruth=gt, ids=ids,
                        dist=np.asarray(dist), k=np.int64(k),
                        metrics_json=np.frombuffer(json.dumps(metrics).encode(), np.uint8),
                        reference_run_json=np.frombuffer(json.dumps(saved).encode(), np.uint8))
    print("reference harness run:", json.dumps(saved["results"], indent=1)[:600])
    print("metrics on the fixture index:", metrics)
    shutil.rmtree(scratch)


if __name__ == "__main__":
    main()
