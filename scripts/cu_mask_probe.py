"""Is a long query's per-expansion latency under load a matter of its own CU or of the memory system?
One query (the longest of a sample) is timed (a) alone, (b) while a 10,000-query batch runs on the SAME CUs, (c) while that
batch runs on a stream masked to CUs 16.. and the query on a stream masked to CUs 0..15 (hipExtStreamCreateWithCUMask).
Two index handles (each has its own scratch and statistics).  usage: python scripts/cu_mask_probe.py [config]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
import numpy as np
import torch
import bench, cphnsw_mi355x

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[cfgname]
class A: workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench"); config = cfgname
path = bench.index_path(A, cfg, cfg["n"])
k = cfg["k"]
Q = bench.make_queries(cfg, cfg["n"], 10000)
big = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); big.load(path)
one = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); one.load(path)
dev = torch.device("cuda", 0)
qb = torch.from_numpy(Q).to(dev)
# the longest query of the batch
big.search_batch_device(qb, k); torch.cuda.synchronize()
ex = big.last_query_expansions(len(Q)) if hasattr(big, "last_query_expansions") else None
if ex is None:
    raise SystemExit("no per-query expansion counts in this build")
li = int(np.argmax(ex))
print(f"longest query: {li}, {int(ex[li])} expansions (mean {float(np.mean(ex)):.0f})", flush=True)
ql = qb[li:li + 1].contiguous()

hip = C.CDLL("libamdhip64.so")
def masked_stream(lo, hi, total=256):
    words = (total + 31) // 32
    m = (C.c_uint32 * words)()
    for cu in range(lo, hi):
        m[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
    if rc != 0:
        raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return s.value

def run(label, s_big, s_one, with_big):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        if with_big:
            big.search_batch_device(qb, k, stream=s_big)
            time.sleep(0.0003)                      # the batch is on the machine when the query starts
        one.search_batch_device(ql, k, stream=s_one)
        torch.cuda.synchronize()
        ts.append(one.last_search_stats()["kernel_us"])
    bs = big.last_search_stats()["kernel_us"] if with_big else None
    print(json.dumps({"case": label, "query_kernel_us_median": float(np.median(ts)), "runs": ts, "batch_kernel_us": bs}), flush=True)

sA = torch.cuda.Stream(); sB = torch.cuda.Stream()
run("alone", None, sB, False)
run("batch on the same CUs (two plain streams)", sA, sB, True)
NE = int(os.environ.get("CPH_EXPRESS_CUS", 16))
mA = masked_stream(NE, 256); mB = masked_stream(0, NE)
run(f"alone on a stream masked to CUs 0..{NE - 1}", None, mB, False)
run(f"batch on CUs {NE}..255, query on CUs 0..{NE - 1}", mA, mB, True)
