"""Two-stream throughput and single-kernel time of the bench batch for a few resident-slot counts
(set_search_params(slots=...)).  usage: slots_sweep.py [config] [slots,slots,...]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rabitq-ann-search_amd"))
import numpy as np, torch
import bench, cphnsw_mi355x
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[cfgname]
class A: workdir = os.environ.get("CPH_BENCH_DIR", "/tmp/cph_bench"); config = cfgname
path = bench.index_path(A, cfg, cfg["n"])
Q = torch.from_numpy(bench.make_queries(cfg, cfg["n"], cfg["nq"])).cuda()
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); ix.load(path)
for slots in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,6144,5120,4096,3072").split(",")]:
    ix.set_search_params(slots=slots, beam_capacity=0)
    ks = []
    for i in range(10):
        ix.search_batch_device(Q, cfg["k"], stream=streams[0]); st = ix.last_search_stats()
        if i >= 3: ks.append(st["kernel_us"])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): ix.search_batch_device(Q, cfg["k"], stream=streams[i & 1])
        ix.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"slots": slots, "kernel_us_mean": float(np.mean(ks)), "pipelined_qps": 20 * cfg["nq"] / dt}), flush=True)
