"""What the estimated-set probe of the gate workloads touches, from the oracle's probe trace (CPU, test infrastructure):
distinct bitmap lines per expansion under the index's own vertex order and under alternatives, and the reuse distance of
the probes that come back "seen" (how many would an exact cache of the last C marked / probed ids certify?).
    python scripts/probe_locality.py <index.idx> [n_queries] [k] [ids_per_line]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402


def trace(oi, q, k, cap=1 << 26):
    out = np.zeros(cap, np.uint32)
    f = oi.o.lib.orc_probe_trace
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_long, np.ctypeslib.ndpointer(np.uint32, flags="C"), C.c_long]
    n = f(oi.h, np.ascontiguousarray(q, np.float32), k, out, cap)
    return out[:min(n, cap)]


def main():
    path = sys.argv[1]
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ipl = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    oi = Oracle().load(path)
    rng = np.random.default_rng(11)
    Q = rng.standard_normal((nq, oi.dim)).astype(np.float32)
    for qi in range(nq):
        t = trace(oi, Q[qi], k)
        head = np.flatnonzero(t & 0x80000000)
        n_exp = len(head)
        ids = t[(t & 0x80000000) == 0]
        exp_of = np.repeat(np.arange(n_exp), np.diff(np.append(head, len(t))) - 1)
        # distinct lines per expansion (ids_per_line ids share a bitmap line)
        line = ids // ipl
        key = exp_of.astype(np.int64) * (1 << 32) + line
        lines_per_exp = len(np.unique(key)) / n_exp
        # the same under a random relabelling (no locality at all)
        perm = rng.permutation(oi.n).astype(np.uint32)
        key_r = exp_of.astype(np.int64) * (1 << 32) + perm[ids] // ipl
        # reuse distance (in expansions) of every probe whose id was probed before
        last = {}
        first = np.zeros(len(ids), bool)
        dist = np.zeros(len(ids), np.int64)
        for j, (v, e) in enumerate(zip(ids.tolist(), exp_of.tolist())):
            p = last.get(v)
            if p is None:
                first[j] = True
            else:
                dist[j] = e - p
            last[v] = e
        seen = ~first
        d = dist[seen]
        qs = [1, 2, 4, 8, 16, 32, 64, 128, 256, 1024]
        cdf = {f"<={x}": round(float((d <= x).mean()), 3) for x in qs}
        print({"query": qi, "expansions": n_exp, "probes": len(ids), "new": int(first.sum()), "seen_frac": round(float(seen.mean()), 3),
               "lines_per_expansion": round(lines_per_exp, 2), "random_order": round(len(np.unique(key_r)) / n_exp, 2),
               "reuse_distance_cdf_of_seen_probes(expansions)": cdf})


if __name__ == "__main__":
    main()
