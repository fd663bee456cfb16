"""Aggregate throughput of T threads calling search() on ONE index handle, ours against the compiled reference on the same
box (the reference: shared lock + GIL released, src/bindings.cpp:146-175; ours: callers coalesced into shared launches).
    python scripts/concurrent_search.py [config] [threads] [calls]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(index, Q, k, T, calls):
    start = threading.Barrier(T + 1)

    def work(t):
        start.wait()
        for c in range(calls):
            index.search(Q[(t * calls + c) % len(Q)], k)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    start.wait()
    t0 = time.perf_counter()
    [x.join() for x in th]
    return T * calls / (time.perf_counter() - t0)


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    cfg = bench.CONFIGS[cfgname]

    class A:
        workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench")
        config = cfgname
    path = bench.index_path(A, cfg, cfg["n"])
    Q = bench.make_queries(cfg, cfg["n"], 2000)
    import cphnsw_mi355x
    ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"])
    ix.load(path)
    out = {"config": cfgname, "k": cfg["k"], "calls_per_thread": calls, "ours_qps": {}, "reference_qps": {}}
    for T in (1, 2, 4, 8, 16, 32) if len(sys.argv) <= 2 else (int(sys.argv[2]),):
        run(ix, Q, cfg["k"], T, 20)
        out["ours_qps"][T] = round(run(ix, Q, cfg["k"], T, calls))
    from oracle_lib import ref_available, ref_module
    if ref_available():
        r = ref_module().CPIndex(cfg["dim"], cfg["bits"])
        r.load(path)
        for T in out["ours_qps"]:
            run(r, Q, cfg["k"], T, 20)
            out["reference_qps"][T] = round(run(r, Q, cfg["k"], T, calls))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
