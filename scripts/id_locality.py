"""How local are neighbour ids on the bench index?  For each vertex: number of distinct 64-B lines
(512 ids) and 128-B lines of a per-query visited bitmap that its 32 neighbours fall into.
Reads the v2 index file directly (layout: tests/golden_util.vertex_layout)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import vertex_layout  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/cph_bench/bench_n1000000_b4.idx"
dim, D, bits = 128, 128, 4
raw = np.memmap(path, dtype=np.uint8, mode="r")
hdr = np.frombuffer(raw[:68].tobytes(), np.uint32)
n = int(hdr[7])
vb, nb_off, codes = vertex_layout(D, bits)
fixed = 68 + 248 + 72 + dim * 4
base = fixed + n * 4 + n * 4 + n * D * 4
ids_off = nb_off + codes + 3 * 128 + 64 + (64 if bits > 1 else 0)
print("n =", n, "vertex bytes", vb, "ids offset", ids_off)
sample = np.random.default_rng(0).choice(n, 20000, replace=False)
l64 = []
l128 = []
span = []
for v in sample:
    o = base + int(v) * vb + ids_off
    ids = np.frombuffer(raw[o:o + 128].tobytes(), np.uint32)
    cnt = int(np.frombuffer(raw[o + 128:o + 132].tobytes(), np.uint32)[0])
    ids = ids[:cnt]
    l64.append(len(np.unique(ids >> 9)))
    l128.append(len(np.unique(ids >> 10)))
    span.append(float(np.abs(ids.astype(np.int64) - int(v)).mean()))
print("distinct 64-B bitmap lines per vertex: mean %.1f  | 128-B lines: mean %.1f | mean |id - own id| %.0f" % (
    np.mean(l64), np.mean(l128), np.mean(span)))
