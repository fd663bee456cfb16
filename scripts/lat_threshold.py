"""Kernel time of mid-sized batches with the default slot policy against half / three quarters of the batch in slots.
(Written when the kernel still had a latency mode -- one wave per query with a next-top prefetch, the default whenever
the batch fit the resident slots: profiles/r2_lat_threshold.md.  The prefetch is gone; scripts/slot_fraction_sweep.py
is the current tool.)  usage: lat_threshold.py [config]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
import numpy as np, torch
import bench, cphnsw_mi355x
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[cfgname]
class A: workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench"); config = cfgname
path = bench.index_path(A, cfg, cfg["n"])
Qall = torch.from_numpy(bench.make_queries(cfg, cfg["n"], 8000)).cuda()
ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); ix.load(path)
for nq in (500, 1000, 2000, 3000, 4000, 5000, 6000):
    row = {"nq": nq}
    for name, slots in (("latency_mode", 0), ("slots_half", max(64, nq // 2)), ("slots_3q", max(64, 3 * nq // 4))):
        ix.set_search_params(slots=slots, beam_capacity=0)
        ks = []
        for i in range(8):
            ix.search_batch_device(Qall[:nq], cfg["k"]); st = ix.last_search_stats()
            if i >= 2: ks.append(st["kernel_us"])
        row[name + "_us"] = float(np.mean(ks))
    print(json.dumps(row), flush=True)
