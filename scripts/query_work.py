"""Distribution of per-query work (vertices expanded) on the bench index and how well cheap
proxies predict it (for scheduling long queries first).  Run on the GPU box after bench.py has
built /tmp/cph_bench/bench_n1000000_b4.idx."""
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
ids, dist = idx.search_batch_device(q, 10)
torch.cuda.synchronize()
e = idx.last_query_expansions(nq).astype(np.float64)
print("expansions: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (e.mean(), *np.percentile(e, [50, 90, 99]), e.max()))
d = dist.cpu().numpy()
eps = np.array([idx.entry_point(Q[i]) for i in range(nq)], np.int64)
epd = np.empty(nq)
for i in range(0, nq, 1000):
    for j in range(i, min(nq, i + 1000)):
        v = idx.get_vectors(int(eps[j]), 1)[0].astype(np.float64)
        epd[j] = ((Q[j].astype(np.float64) - v) ** 2).sum()
proxies = {
    "entry distance": epd,
    "result d[0]": d[:, 0],
    "result d[9]": d[:, 9],
    "|q|^2": (Q.astype(np.float64) ** 2).sum(1),
}
for name, p in proxies.items():
    r = np.corrcoef(np.argsort(np.argsort(p)), np.argsort(np.argsort(e)))[0, 1]
    print(f"rank correlation with {name}: {r:.3f}")

# what a schedule is worth: simulate a greedy work queue with `slots` workers
def makespan(order, slots=5000):
    import heapq
    h = [0.0] * slots
    heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + e[i])
    return max(h)

base = makespan(np.arange(nq))
print("greedy queue makespan (expansions): given order %.0f | longest-first %.0f | lower bound %.0f" % (
    base, makespan(np.argsort(-e)), max(e.max(), e.sum() / 5000)))
for name, p in proxies.items():
    print(f"  ordered by {name} desc: {makespan(np.argsort(-p)):.0f}   asc: {makespan(np.argsort(p)):.0f}")
