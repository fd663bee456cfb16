"""dev tool: per-pass time of the stream kernel vs number of back-to-back passes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rabitq-ann-search_amd"))
import cphnsw_mi355x
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
st = cphnsw_mi355x.FastScanStream(128, 4, nb, seed=4)
for reps in (1, 5, 20, 100, 400, 20, 1):
    ms, _ = st.run(reps)
    print(reps, round(ms, 4), "ms/pass", round(nb * 32 * 84 / ms / 1e6, 1), "GB/s alg")
