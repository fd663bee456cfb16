"""Kernel time of a batch of nq queries against the fraction of it that gets a resident slot (1.0 = one wave per
query = the kernel's latency mode with the next-top prefetch).  usage: slot_fraction_sweep.py [nq,nq,..] [frac,frac,..]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
import numpy as np, torch
import bench, cphnsw_mi355x
cfg = bench.CONFIGS["c2"]
class A: workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench"); config = "c2"
path = bench.index_path(A, cfg, cfg["n"])
Qall = torch.from_numpy(bench.make_queries(cfg, cfg["n"], 8000)).cuda()
ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); ix.load(path)
for nq in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1000,2000,4000,6000,8000').split(',')]:
    row = {"nq": nq}
    for frac in [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '0.5,0.625,0.75,0.875,0.95').split(',')]:
        ix.set_search_params(slots=min(6144, max(16, int(nq * frac))), beam_capacity=0)   # frac 1.0 = latency mode
        ks = []
        for i in range(8):
            ix.search_batch_device(Qall[:nq], cfg["k"]); st = ix.last_search_stats()
            if i >= 2: ks.append(st["kernel_us"])
        row[str(frac)] = round(float(np.mean(ks)), 1)
    print(json.dumps(row), flush=True)
