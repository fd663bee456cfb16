"""Histogram of the neighbour-list lengths (`count`) of a v2 index file:  python scripts/count_hist.py <file.idx> dim bits"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from golden_util import vertex_layout_full

path, dim, bits = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
D = 1 << (dim - 1).bit_length()
hdr = np.fromfile(path, np.uint8, 64)
n = int(hdr[28:36].view(np.uint64)[0])
vb, nb_off, _, cnt_off = vertex_layout_full(D, bits)
base = 68 + 248 + 72 + dim * 4 + n * 4 + n * 4 + n * D * 4
mm = np.memmap(path, np.uint8, "r", offset=base, shape=(n, vb))
cnt = np.ascontiguousarray(mm[:, nb_off + cnt_off:nb_off + cnt_off + 4]).view(np.uint32).ravel()
h = np.bincount(cnt, minlength=33)
print({"n": n, "count_hist": {int(i): int(c) for i, c in enumerate(h) if c}, "lists_with_tail": int((cnt % 8 != 0).sum())})
