"""One-off assurance run on the GPU box: every query of the bench batch (10,000) against the
compiled reference (oracle/_ref) on the 1M bench index at k = 10 and k = 100, ids and distances
bit for bit.  Needs /tmp/cph_bench/bench_n1000000_b4.idx (bench.py builds it)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cphnsw_mi355x import CPIndex  # noqa: E402
from oracle_lib import ref_module  # noqa: E402

path = "/tmp/cph_bench/bench_n1000000_b4.idx"
_, Q = bench.make_data(1000000, 10000, need_base=False)
g = CPIndex(128, 4)
g.load(path)
r = ref_module().CPIndex(128, 4)
r.load(path)
for k in (10, 100):
    t = time.time()
    rid, rd = r.search_batch(Q, k)
    tr = time.time() - t
    gid, gd = g.search_batch(Q, k)
    print(f"k={k}: ids identical {np.array_equal(rid, gid)}, distances bit-identical {rd.tobytes() == gd.tobytes()}"
          f"  (reference {len(Q) / tr:.0f} QPS on {os.cpu_count()} threads)", flush=True)
