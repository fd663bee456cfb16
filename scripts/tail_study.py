"""How much of a 10k-query batch is the tail of its longest queries?  Kernel time of the bench batch with
the p% longest queries (by measured expansions) removed.  Run on the GPU box after bench.py built the index."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from cphnsw_mi355x import CPIndex  # noqa: E402

nq = 10000
_, Q = bench.make_data(1000000, nq, need_base=False)
idx = CPIndex(dim=128, bits=4)
idx.load("/tmp/cph_bench/bench_n1000000_b4.idx")
q = torch.from_numpy(Q).cuda()
idx.search_batch_device(q, 10)
e = idx.last_query_expansions(nq).astype(np.int64)
order = np.argsort(-e)
for drop in (0.0, 0.01, 0.05, 0.2, 0.5):
    keep = np.sort(order[int(drop * nq):])
    qs = q[torch.from_numpy(keep).cuda()].contiguous()
    best = 1e9
    for _ in range(5):
        idx.search_batch_device(qs, 10)
        torch.cuda.synchronize()
        best = min(best, idx.last_search_stats()["kernel_us"])
    tot = int(e[keep].sum())
    print(f"drop longest {drop:4.0%}: {len(keep):5d} queries, {tot:8d} expansions, longest {int(e[keep].max()):4d}, "
          f"kernel {best:5d} us = {best * 1e3 / tot:6.3f} ns per expansion")
