"""Shared definitions for the golden fixtures (datasets, calibration patches, loaders).

Pure data plumbing: no reference code.  ``make_golden.py`` (authoring container) and the
tests use the same seeded generators and the same byte patches, so a fixture index file
plus a variant name identifies exactly the bytes the reference searched.
"""
import gzip
import os
import struct
import tempfile

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KS = (1, 10, 20, 100)
CALIB_OFF = 68  # CalibrationSnapshot offset in the v2 file

DATASETS = {
    # name: n, dim, D (padded), bit-widths, kind, seed, variants
    "g128": dict(n=400, dim=128, D=128, bits=(1, 2, 4), kind="gauss", seed=101,
                 variants=("plain", "gamma", "affine", "bignop", "shortcount")),
    "sift96": dict(n=400, dim=96, D=128, bits=(4,), kind="sift", seed=102,
                   variants=("plain", "gamma")),
    "g16": dict(n=300, dim=10, D=16, bits=(1, 2, 4), kind="gauss", seed=103,
                variants=("plain", "gamma_tight", "shortcount")),
    "g1024": dict(n=160, dim=960, D=1024, bits=(2,), kind="gauss", seed=104,
                  variants=("plain", "affine", "shortcount")),
}
NQ = 24

# calibration patches (field offsets inside CalibrationSnapshot, api/hnsw_index.hpp:33-58)
_F = {"affine_a": 0, "affine_b": 4, "ip_qo_floor": 8, "gamma_max": 84, "gamma_beta": 88,
      "search_gamma": 240}
VARIANTS = {
    "plain": {},
    "gamma": dict(search_gamma=1.1, gamma_max=1.6, gamma_beta=0.7, gamma_warmup=3),
    "gamma_tight": dict(search_gamma=1.0, gamma_max=1.0, gamma_beta=0.0, gamma_warmup=1),
    "affine": dict(affine_a=0.93, affine_b=0.02, ip_qo_floor=0.55, search_gamma=1.3,
                   gamma_max=2.5, gamma_beta=1.0, gamma_warmup=8, slack=[-0.05, 0.0, 0.02]),
    # every 3rd vertex gets nop=500 for all 32 neighbours -> stage-2 skip branch
    "bignop": dict(bignop=(3, 500.0)),
    # every 3rd vertex' neighbour list is cut short (count patched; the slots behind it keep their stale ids, as a
    # re-pruned list does, graph/graph_refinement.hpp:46-47): the scalar tails of the epilogues for count % 8 != 0
    # (distance/fastscan_kernel.hpp:174-193, :324-345), batch_count < 32 (search/rabitq_search.hpp:152-154) and the
    # empty list (:137).  With gamma finite so that the estimates of the tail lanes decide something.
    "shortcount": dict(shortcount=(3, (0, 5, 13, 29, 31)), search_gamma=1.2, gamma_max=1.8, gamma_beta=0.5,
                       gamma_warmup=4),
}
SHORT_COUNTS = (5, 13, 29)   # F/ vectors: counts with a scalar tail (count % 8 != 0)


def sift_like(rng, n, dim, ncl=40):
    cent = rng.gamma(2, 15, (ncl, dim))
    X = cent[rng.integers(0, ncl, n)] + rng.normal(0, 12, (n, dim))
    return np.clip(np.round(X), 0, 218).astype(np.float32)


def make_dataset(name):
    s = DATASETS[name]
    rng = np.random.default_rng(s["seed"])
    if s["kind"] == "sift":
        X = sift_like(rng, s["n"], s["dim"])
        Q = sift_like(rng, NQ, s["dim"])
    else:
        X = rng.standard_normal((s["n"], s["dim"])).astype(np.float32)
        Q = rng.standard_normal((NQ, s["dim"])).astype(np.float32)
    Q[0] = X[7]  # exact hit: exercises the dist_qp_sq < 1e-12 early-outs
    return X, Q


def vertex_layout(D, bits):
    """(vertex_bytes, nb_off, nop_off) of VertexSearchData<D,32,bits>."""
    return vertex_layout_full(D, bits)[:3]


def vertex_layout_full(D, bits):
    """(vertex_bytes, nb_off, nop_off, count_off): count_off = offset of `count` inside the neighbour block."""
    def up(x, a):
        return (x + a - 1) // a * a
    words = (D + 63) // 64
    code = up(up(bits * words * 8, 64) + 8, 64)
    codes = bits * up((D // 8) * 32, 64)
    o = codes + 3 * 128 + 64 + (64 if bits > 1 else 0) + 128 + 4
    return code + up(o, 64), code, codes, o - 4


def apply_variant(data, vname, spec, bits):
    return apply_patch(data, VARIANTS[vname], spec, bits)


def apply_patch(data, kw, spec=None, bits=None):
    """Calibration / graph patches on the bytes of a v2 index file (spec and bits only for the graph patches)."""
    b = bytearray(data)
    for k, v in kw.items():
        if k in _F:
            struct.pack_into("<f", b, CALIB_OFF + _F[k], v)
        elif k == "gamma_warmup":
            struct.pack_into("<Q", b, CALIB_OFF + 96, v)
        elif k == "slack":
            for i, x in enumerate(v):
                struct.pack_into("<f", b, CALIB_OFF + 108 + 4 * i, x)
            struct.pack_into("<i", b, CALIB_OFF + 236, len(v))
        elif k == "bignop":
            every, val = v
            n, dim, D = spec["n"], spec["dim"], spec["D"]
            vb, nb_off, nop_off = vertex_layout(D, bits)
            base = 68 + 248 + 72 + dim * 4 + n * 4 + n * 4 + n * D * 4
            blob = np.full(32, val, np.float32).tobytes()
            for vtx in range(0, n, every):
                off = base + vtx * vb + nb_off + nop_off
                b[off:off + 128] = blob
        elif k == "shortcount":
            every, counts = v
            n, dim, D = spec["n"], spec["dim"], spec["D"]
            vb, nb_off, _, cnt_off = vertex_layout_full(D, bits)
            base = 68 + 248 + 72 + dim * 4 + n * 4 + n * 4 + n * D * 4
            for j, vtx in enumerate(range(1, n, every)):
                off = base + vtx * vb + nb_off + cnt_off
                old, = struct.unpack_from("<I", b, off)
                struct.pack_into("<I", b, off, min(old, counts[j % len(counts)]))
    return bytes(b)


_TMP = None


def fixture_path(name, bits, variant="plain"):
    """Materialise a (possibly patched) fixture index as a real file; returns its path."""
    global _TMP
    if _TMP is None:
        _TMP = tempfile.mkdtemp(prefix="cph_golden_")
    p = os.path.join(_TMP, f"idx_{name}_b{bits}_{variant}.idx")
    if not os.path.exists(p):
        with gzip.open(os.path.join(GOLDEN_DIR, f"idx_{name}_b{bits}.idx.gz"), "rb") as g:
            data = g.read()
        with open(p, "wb") as f:
            f.write(apply_variant(data, variant, DATASETS[name], bits))
    return p


_G = None


def golden():
    global _G
    if _G is None:
        _G = np.load(os.path.join(GOLDEN_DIR, "golden.npz"))
    return _G
