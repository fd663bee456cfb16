"""A/B timing tool on the cached bench index: kernel time of serialised batches and two-stream throughput, one JSON
line per repetition (second argument: a comma-separated list, one repetition per entry).  CPH_LIB_PATH selects a
variant library, so two builds can be compared on the same box:
    CPH_LIB_PATH=$PWD/build/variants/a.so python scripts/lat_sweep.py c2 0,0"""
import os, sys, time, json
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
for lat in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,256,512,1024,2048,4096").split(",")]:
    ix = cphnsw_mi355x.CPIndex(cfg["dim"], cfg["bits"]); ix.load(path)
    ks = []
    for i in range(13):
        ix.search_batch_device(Q, cfg["k"], stream=streams[0]); st = ix.last_search_stats()
        if i >= 3: ks.append(st["kernel_us"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): ix.search_batch_device(Q, cfg["k"], stream=streams[i & 1])
    ix.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"rep": lat, "kernel_us_mean": float(np.mean(ks)), "kernel_us_min": int(min(ks)), "pipelined_qps": 20 * cfg["nq"] / dt}), flush=True)
    del ix
