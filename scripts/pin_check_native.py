"""Limit of the pinning, checked: do the goldens (made with the reference compiled -march=x86-64-v3, oracle/Makefile) also
hold for the reference as its own CMake compiles it, -march=native, on a host with AVX-512?  Authoring container only.
    g++ -O3 -std=c++17 -march=native -mavx2 -mfma -fopenmp -fPIC -w -shared -I/root/reference/include oracle/ref_hooks.cpp -o /tmp/refnative/libcph_refhooks.so
    g++ ... $(python3 -m pybind11 --includes) /root/reference/src/bindings.cpp -o /tmp/refnative/_core$(python3-config --extension-suffix)
    python scripts/pin_check_native.py /tmp/refnative"""
import ctypes as C
import glob
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from golden_util import DATASETS, KS, SHORT_COUNTS, fixture_path, golden  # noqa: E402

d = sys.argv[1]
g = golden()
hooks = oracle_lib._Hooks(C.CDLL(os.path.join(d, "libcph_refhooks.so")), "ref_")
spec = importlib.util.spec_from_file_location("_core", glob.glob(os.path.join(d, "_core*.so"))[0])
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
bad = 0
n_f = n_s = 0
for D in (16, 128, 1024):
    for bits in (1, 2, 4):
        k = f"F/{D}/b{bits}"
        for i in range(g[f"{k}/planes"].shape[0]):
            for a, qp in enumerate(g[f"{k}/qps"]):
                for c, dqp in enumerate(g[f"{k}/dqps"]):
                    for cnt in (32,) + tuple(SHORT_COUNTS):
                        args = (g[f"{k}/nop"][i], g[f"{k}/ipqo"][i], g[f"{k}/ipcp"][i], g[f"{k}/pop"][i])
                        if bits == 1:
                            e, lo = hooks.convert_1bit(D, qp, g[f"{k}/sums"][i], *args, dqp, cnt)
                            lo1 = lo
                        else:
                            lo1 = hooks.convert_msb(D, bits, qp, g[f"{k}/msb2"][i], *args, dqp, cnt)
                            e, lo = hooks.convert_nbit(D, bits, qp, g[f"{k}/sums"][i], g[f"{k}/msb"][i], *args, g[f"{k}/wpop"][i], dqp, cnt)
                        sub = "" if cnt == 32 else f"/c{cnt}"
                        for got, nm in ((e, "est"), (lo, "lower"), (lo1, "lower1")):
                            n_f += 1
                            bad += got[:cnt].tobytes() != g[f"{k}{sub}/{nm}"][a, c, i, :cnt].tobytes()
for D, dim in ((16, 10), (128, 128), (128, 96), (1024, 960)):
    q = g[f"E/{D}/{dim}/q"]
    for i in range(len(q)):
        lut, co, rot = hooks.encode_query(q[i], D)
        n_f += 1
        bad += not (np.array_equal(lut, g[f"E/{D}/{dim}/lut"][i]) and co.tobytes() == g[f"E/{D}/{dim}/coeffs"][i].tobytes())
for name, spec_ in DATASETS.items():
    for bits in spec_["bits"]:
        for v in spec_["variants"]:
            ix = mod.CPIndex(spec_["dim"], bits)
            ix.load(fixture_path(name, bits, v))
            for k in KS:
                ids, dd = ix.search_batch(g[f"Q/{name}"], k)
                n_s += 1
                bad += not (np.array_equal(ids, g[f"S/{name}/b{bits}/{v}/k{k}/ids"]) and dd.tobytes() == g[f"S/{name}/b{bits}/{v}/k{k}/d"].tobytes())
# the data-side encoder (the vectoriser-sensitive loops of DESIGN.md section 5: fused / unfused boundary, dim % 4 remainder)
gb = np.load(os.path.join(ROOT, "tests", "golden", "golden_build.npz"))
n_e = 0
for key in sorted({k.rsplit("/", 1)[0] for k in gb.files if k.startswith("ENC/")}):
    _, dim, D, b = key.split("/")
    for c in range(len(gb[f"{key}/parent"])):
        v, a, s_ = hooks.encode_edges(gb[f"{key}/parent"][c], gb[f"{key}/nbrs"][c], int(D), int(b[1:]))
        n_e += 1
        bad += not (np.array_equal(v, gb[f"{key}/values"][c]) and a.tobytes() == gb[f"{key}/aux"][c].tobytes() and np.array_equal(s_, gb[f"{key}/pops"][c]))
print({"epilogue_and_encoder_vectors_checked": n_f, "search_outputs_checked": n_s, "edge_encoder_cases_checked": n_e, "mismatches": bad})
