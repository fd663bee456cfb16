"""Times the MFMA brute-force kNN (csrc/device_knn.h) and checks it against a float64 brute force on a
sample of rows.  usage: python scripts/knn_time.py [n] [dim]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rabitq-ann-search_amd"))
import cphnsw_mi355x

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(1)
cent = rng.gamma(2.0, 15.0, (max(10, n // 1000), dim))
X = np.clip(np.round(cent[rng.integers(0, len(cent), n)] + rng.normal(0, 12.0, (n, dim))), 0, 218).astype(np.float32)
cphnsw_mi355x.knn_bruteforce(X[:4096])          # warm-up (library load, clocks)
t0 = time.time()
ids, d = cphnsw_mi355x.knn_bruteforce(X)
dt = time.time() - t0
D = 1 << (dim - 1).bit_length()
print(f"n={n} dim={dim} (D={D}): {dt:.2f} s wall incl. H2D/D2H  ->  {2.0 * n * n * D / dt / 1e12:.1f} TFLOP/s (padded D)")
rows = rng.integers(0, n, 64)
for r in rows:
    dd = ((X.astype(np.float64) - X[r].astype(np.float64)) ** 2).sum(1)
    dd[r] = np.inf
    want = np.sort(dd)[:32]
    assert np.allclose(d[r], want, rtol=1e-4, atol=1e-3), (r, d[r][:4], want[:4])
    assert np.allclose(dd[ids[r].astype(np.int64)], want, rtol=1e-4, atol=1e-3)
print("sample of 64 rows matches float64 brute force")
