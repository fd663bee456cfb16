"""Dataset readers for the benchmark harness (counterpart of the reference's cphnsw/datasets.py:
same directory layout, same returned dict).  *.fvecs / *.ivecs are the TEXMEX formats (each record:
int32 length d, then d values); the npy datasets are three arrays in one directory."""
from pathlib import Path

import numpy as np

FVECS_DATASETS = {
    "sift1m": ("sift_base.fvecs", "sift_query.fvecs", "sift_groundtruth.ivecs"),
    "gist1m": ("gist_base.fvecs", "gist_query.fvecs", "gist_groundtruth.ivecs"),
}
NPY_DATASETS = ("msmarco10m", "openai1536")
ALL_DATASETS = list(FVECS_DATASETS) + sorted(NPY_DATASETS)


def read_vecs(path, dtype):
    """One TEXMEX file -> (n, d) array of `dtype` (float32 for fvecs, int32 for ivecs)."""
    raw = np.fromfile(path, dtype=np.int32)
    if raw.size == 0:
        return np.zeros((0, 0), dtype)
    d = int(raw[0])
    rec = raw.reshape(-1, d + 1)
    if not (rec[:, 0] == d).all():
        raise ValueError(f"{path}: inconsistent record lengths")
    return np.ascontiguousarray(rec[:, 1:]).view(dtype)


def write_vecs(path, arr):
    """Inverse of read_vecs (used by the tests to fabricate small datasets)."""
    arr = np.ascontiguousarray(arr)
    n, d = arr.shape
    rec = np.empty((n, d + 1), np.int32)
    rec[:, 0] = d
    rec[:, 1:] = arr.view(np.int32)
    rec.tofile(path)


def load_dataset(name, base_dir):
    root = Path(base_dir) / name
    if name in FVECS_DATASETS:
        b, q, g = FVECS_DATASETS[name]
        base = read_vecs(root / b, np.float32)
        queries = read_vecs(root / q, np.float32)
        gt = read_vecs(root / g, np.int32)
    else:
        base = np.load(root / "base.npy").astype(np.float32)
        queries = np.load(root / "queries.npy").astype(np.float32)
        gt = np.load(root / "groundtruth.npy").astype(np.int32)
    return {"base": base, "queries": queries, "groundtruth": gt, "dim": int(base.shape[1])}
