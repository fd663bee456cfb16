#!/usr/bin/env python3
"""Golden vectors for the construction side (authoring container only; needs oracle/_ref).

    python tests/golden/make_golden_build.py   ->   tests/golden/golden_build.npz

Everything written is data: seeded inputs and what the REAL reference computed for them.
    ENC/<dim>/<D>/b<bits>/{parent,nbrs,values,aux,pops}   data-side edge encoder
        (encoder/rabitq_encoder.hpp:138-181, 287-323, 371-467 through oracle/ref_hooks.cpp:ref_encode_edges)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import RefHooks  # noqa: E402

ENC_SHAPES = ((10, 16), (50, 64), (96, 128), (128, 128), (300, 512), (960, 1024))


def enc_inputs(dim, bits, seed):
    """Half Gaussian at a random scale, half SIFT-like integer data with an exact duplicate (nop = 0)."""
    rng = np.random.default_rng([7, dim, bits, seed])
    if seed % 2 == 0:
        sc = 10.0 ** rng.uniform(-2, 2)
        p = (sc * rng.standard_normal(dim)).astype(np.float32)
        nb = (p + sc * rng.uniform(0.05, 1.5) * rng.standard_normal((32, dim))).astype(np.float32)
    else:
        p = np.round(rng.gamma(2, 15, dim)).astype(np.float32)
        nb = np.clip(np.round(p + rng.normal(0, 12, (32, dim))), 0, 218).astype(np.float32)
        nb[5] = p
    return p, nb


def main():
    r = RefHooks()
    out = {}
    for dim, D in ENC_SHAPES:
        for bits in (1, 2, 4):
            P, N, V, A, S = [], [], [], [], []
            for seed in range(2):
                p, nb = enc_inputs(dim, bits, seed)
                v, a, s = r.encode_edges(p, nb, D, bits)
                P.append(p), N.append(nb), V.append(v), A.append(a), S.append(s)
            key = f"ENC/{dim}/{D}/b{bits}"
            out[f"{key}/parent"], out[f"{key}/nbrs"] = np.stack(P), np.stack(N)
            out[f"{key}/values"], out[f"{key}/aux"], out[f"{key}/pops"] = np.stack(V), np.stack(A), np.stack(S)
    np.savez_compressed(os.path.join(HERE, "golden_build.npz"), **out)
    print("wrote golden_build.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
