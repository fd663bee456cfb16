"""Builds an index with the GPU builder and prints the stage times (CPH_BUILD_VERBOSE=1).
usage: python scripts/time_build.py [config c2|c3|c4|recall] [n] [--check N]   (--check: N queries vs the compiled reference)"""
import os, sys, time
os.environ.setdefault("CPH_BUILD_VERBOSE", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, cphnsw_mi355x
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[cfgname]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else cfg["n"]
check = int(sys.argv[sys.argv.index("--check") + 1]) if "--check" in sys.argv else 0
X = bench.make_base(cfg, n)
t0 = time.time()
ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"])
ix.build(X)
ix.finalize()
print(f"build+finalize {cfgname} n={n} dim={cfg['dim']} bits={cfg['bits']}: {time.time() - t0:.1f} s", flush=True)
Q = bench.make_queries(cfg, n, max(check, 1000))
ids, d = ix.search_batch(Q[:1000], cfg["k"])
print("search stats", ix.last_search_stats())
gt = bench.ground_truth(X, Q[:1000], 0)
print("recall@10 dedup k=%d: %.4f" % (cfg["k"], bench.recall_at_10(ids, d, gt, True)))
if check:
    from oracle_lib import ref_available, ref_module
    p = f"/tmp/time_build_{cfgname}_{n}.idx"
    t0 = time.time(); ix.save(p); print(f"save {time.time() - t0:.1f} s")
    if ref_available():
        r = ref_module().CPIndex(cfg["dim"], cfg["bits"]); r.load(p)
        rid, rd = r.search_batch(Q[:check], cfg["k"]); gid, gd = ix.search_batch(Q[:check], cfg["k"])
        print("reference loads our file; ids identical:", np.array_equal(rid, gid), "distances bit-identical:", rd.tobytes() == gd.tobytes())
    os.remove(p)
