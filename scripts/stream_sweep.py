"""Streaming FastScan sweep (dev tool): prints dist/s and GB/s for a (D, bits) list."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rabitq-ann-search_amd"))
import cphnsw_mi355x
cfgs = [tuple(map(int, c.split(","))) for c in (sys.argv[1:] or ["128,4,1000000"])]
for D, bits, nb in cfgs:
    st = cphnsw_mi355x.FastScanStream(D, bits, nb, seed=4)
    st.run(200 if D <= 128 else 30)      # the chip reaches its steady clock
    ms, _ = st.run(100 if D <= 128 else 20)
    bpd = D * bits // 8 + (18 if bits == 1 else 20)
    print(json.dumps({"D": D, "bits": bits, "blocks": nb, "ms": round(ms, 4), "Gdist_s": round(nb * 32 / ms / 1e6, 2),
                      "alg_GBs": round(nb * 32 * bpd / ms / 1e6, 1), "moved_GBs": round(nb * st.block_bytes / ms / 1e6, 1),
                      "frac_of_8TBs": round(nb * 32 * bpd / ms / 1e6 / 8000.0, 4), "pair_kernel": os.environ.get("CPH_STREAM_PAIR", "1")}))
    st.close()
