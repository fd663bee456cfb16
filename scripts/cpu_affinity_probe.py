"""What CPUs does a process on the GPU box get, and does initialising torch / HIP change the calling thread's mask?"""
import os
def show(tag):
    st = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith(("Cpus_allowed_list", "Threads"))}
    print(tag, "affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], st, flush=True)
show("start")
print("cpu_count", os.cpu_count(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
for p in ("/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset.cpus"):
    if os.path.exists(p):
        print(p, open(p).read().strip()[:200])
import numpy  # noqa
show("after numpy")
import torch
show("after import torch")
torch.cuda.init(); torch.zeros(1, device="cuda")
show("after torch.cuda init")
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rabitq-ann-search_amd"))
import cphnsw_mi355x
s = cphnsw_mi355x.FastScanStream(128, 4, 64); s.run(1); s.close()
show("after our library")
