import sys, time, os, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from oracle_lib import ref_module
import bench
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
n = int(sys.argv[1])
X, Q = bench.make_data(n, 1000)
m = ref_module()
idx = m.CPIndex(128, 4)
t=time.time(); idx.build(X); print("build", time.time()-t, flush=True)
t=time.time(); idx.finalize(); print("finalize", time.time()-t, flush=True)
t=time.time(); idx.search_batch(Q, 10); print("search 1000q", time.time()-t, flush=True)
