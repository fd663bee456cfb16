"""Latency of the reference's one-query entry point, CPIndex.search (src/bindings.cpp:138-175), on the cached bench
index: host numpy in, numpy out, one call at a time (latency mode of the search kernel: next-top prefetch, one wave).
Prints one JSON line: median / p90 in microseconds plus the same for a 32-query search_batch."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
import numpy as np
import bench, cphnsw_mi355x

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[cfgname]
class A: workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench"); config = cfgname
path = bench.index_path(A, cfg, cfg["n"])
Q = bench.make_queries(cfg, cfg["n"], 1000)
ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); ix.load(path)
k = cfg["k"]
for q in Q[:20]:
    ix.search(q, k)
one = []
for q in Q[20:520]:
    t0 = time.perf_counter(); ix.search(q, k); one.append(time.perf_counter() - t0)
b32 = []
for i in range(20):
    t0 = time.perf_counter(); ix.search_batch(Q[32 * i:32 * i + 32], k); b32.append(time.perf_counter() - t0)
st = ix.last_search_stats()
print(json.dumps({"config": cfgname, "search_us_median": round(1e6 * float(np.median(one)), 1),
                  "search_us_p90": round(1e6 * float(np.percentile(one, 90)), 1),
                  "batch32_us_median": round(1e6 * float(np.median(b32)), 1),
                  "kernel_us_last_batch32": st["kernel_us"]}))
