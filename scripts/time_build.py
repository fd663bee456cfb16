"""dev tool: build an index of n x dim with our builder (CPH_BUILD_VERBOSE=1 for stage timings), then
report search throughput.   python scripts/time_build.py N [BITS] [DIM] [KIND]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd")); sys.path.insert(0, ROOT)
import numpy as np, bench, cphnsw_mi355x
n = int(sys.argv[1]); bits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
kind = sys.argv[4] if len(sys.argv) > 4 else "sift"
nq = 10000 if dim <= 256 else 2000
rng = np.random.default_rng(2)
if kind == "sift" and dim == 128:
    X, Q = bench.make_data(n, nq)
elif kind == "gauss":
    X = rng.standard_normal((n, dim)).astype(np.float32); Q = rng.standard_normal((nq, dim)).astype(np.float32)
else:  # GIST-like: U[0,1) clusters
    cent = rng.random((500, dim)); X = (cent[rng.integers(0, 500, n)] + rng.normal(0, 0.05, (n, dim))).astype(np.float32)
    Q = (cent[rng.integers(0, 500, nq)] + rng.normal(0, 0.05, (nq, dim))).astype(np.float32)
ix = cphnsw_mi355x.CPIndex(dim, bits)
t = time.time(); ix.build(X); ix.finalize(); bt = time.time() - t
for k in (10,):
    ids, d = ix.search_batch(Q, k)
    t = time.time(); ids, d = ix.search_batch(Q, k); dt = time.time() - t
    st = ix.last_search_stats()
    print({"n": n, "dim": dim, "bits": bits, "kind": kind, "build_s": round(bt, 1), "k": k, "nq": nq, "qps": round(len(Q) / dt),
           "exp_per_q": round(st["expansions"] / len(Q)), "exact_per_q": round(st["exact_l2"] / len(Q)),
           "kernel_ms": st["kernel_us"] / 1e3}, flush=True)
