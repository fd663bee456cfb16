"""dev tool: PCIe-inclusive rate of search_batch (host numpy in, host numpy out) on the bench index."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd")); sys.path.insert(0, ROOT)
import numpy as np, bench, cphnsw_mi355x
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X, Q = bench.make_data(n, 10000)
p = f"/tmp/cph_bench/bench_n{n}_b4.idx"
ix = cphnsw_mi355x.CPIndex(128, 4); ix.load(p)
for _ in range(3): ix.search_batch(Q, 10)
t = time.perf_counter()
for _ in range(10): ix.search_batch(Q, 10)
dt = (time.perf_counter() - t) / 10
print("PCIe-inclusive search_batch: %.2f ms/batch, %.0f QPS" % (dt * 1e3, len(Q) / dt))
