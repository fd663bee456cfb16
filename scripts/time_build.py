"""dev tool: time our builder at size n (CPH_BUILD_VERBOSE stage timings on stderr) and report recall"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd")); sys.path.insert(0, ROOT)
import numpy as np, bench, cphnsw_mi355x
n = int(sys.argv[1]); bits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
X, Q = bench.make_data(n, 10000)
ix = cphnsw_mi355x.CPIndex(128, bits)
t = time.time(); ix.build(X); ix.finalize(); print("build+finalize %.1f s" % (time.time() - t), flush=True)
for k in (10, 20):
    ids, d = ix.search_batch(Q, k); st = ix.last_search_stats()
    t = time.time(); ids, d = ix.search_batch(Q, k); dt = time.time() - t
    print("k=%d  %.0f QPS  exp/q %.0f  exact/q %.0f  kernel %.2f ms" % (k, len(Q) / dt, st["expansions"] / len(Q), st["exact_l2"] / len(Q), st["kernel_us"] / 1e3))
