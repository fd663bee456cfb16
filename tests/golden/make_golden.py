#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference (authoring container only).

Needs oracle/_ref (``make -C oracle ref``), i.e. /root/reference.  Everything written here
is data: index files saved by the reference's own ``CPIndex.save`` (gzip-compressed), the
seeded inputs, and the outputs the reference produced for them.

    python tests/golden/make_golden.py            # re-uses the committed idx_*.idx.gz files
    python tests/golden/make_golden.py --rebuild  # builds new ones (the build is not reproducible across
                                                  # thread counts, SURVEY F6: every S/ vector changes with it)

Outputs (tests/golden/):
    idx_<name>_b<bits>.idx.gz   reference-built v2 index files
    golden.npz                  inputs + expected outputs (see keys below)

Key scheme in golden.npz:
    Q/<name>                              queries for dataset <name>
    S/<name>/b<bits>/<variant>/k<k>/ids   reference search_batch ids   (int64 [nq,k])
    S/<name>/b<bits>/<variant>/k<k>/d     reference search_batch dists (float32 [nq,k])
    E/<D>/<dim>/{q,lut,coeffs,rot}        query-encoder vectors
    F/<D>/b<bits>/...                     FastScan block vectors (sums + fp32 epilogues)
    F/<D>/b<bits>/c<count>/{est,lower,lower1}  the same epilogues called with count = 5, 13, 29: the scalar tails
                                          for count % 8 != 0 (lanes >= count are zero)
    X/<D>/{a,b,dot,l2}                    exact arithmetic vectors
Variants are calibration/graph patches applied to the index file bytes (tests apply the
same patch with tests/golden_util.py), so the fixture set stays small while still covering
the gamma-termination, DABS, gamma-adaptation, affine/floor and stage-2-skip branches that
the reference's own calibration never reaches.
"""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import DATASETS, VARIANTS, KS, SHORT_COUNTS, make_dataset, apply_variant  # noqa: E402
from oracle_lib import RefHooks, ref_module  # noqa: E402


PC8 = np.array([bin(x).count("1") for x in range(256)], np.int64)


def main():
    m = ref_module()
    r = RefHooks()
    out = {}
    tmp = "/tmp/golden_build"
    os.makedirs(tmp, exist_ok=True)

    for name, spec in DATASETS.items():
        X, Q = make_dataset(name)
        out[f"Q/{name}"] = Q
        for bits in spec["bits"]:
            gz = os.path.join(HERE, f"idx_{name}_b{bits}.idx.gz")
            if os.path.exists(gz) and "--rebuild" not in sys.argv:
                with gzip.open(gz, "rb") as g:
                    data = g.read()
            else:
                idx = m.CPIndex(spec["dim"], bits)
                idx.build(X)
                idx.finalize()
                raw = os.path.join(tmp, f"idx_{name}_b{bits}.idx")
                idx.save(raw)
                data = open(raw, "rb").read()
                with gzip.GzipFile(gz, "wb", compresslevel=9, mtime=0) as g:
                    g.write(data)
            for vname in spec["variants"]:
                patched = apply_variant(data, vname, spec, bits)
                p = os.path.join(tmp, f"idx_{name}_b{bits}_{vname}.idx")
                open(p, "wb").write(patched)
                li = m.CPIndex(spec["dim"], bits)
                li.load(p)
                for k in KS:
                    ids, d = li.search_batch(Q, k)
                    out[f"S/{name}/b{bits}/{vname}/k{k}/ids"] = ids
                    out[f"S/{name}/b{bits}/{vname}/k{k}/d"] = d
                # single-query API (unpadded) for the first 4 queries at k=10
                for qi in range(4):
                    ids, d = li.search(Q[qi], 10)
                    out[f"S1/{name}/b{bits}/{vname}/q{qi}/ids"] = ids
                    out[f"S1/{name}/b{bits}/{vname}/q{qi}/d"] = d
            print("built", name, bits, len(data))

    rng = np.random.default_rng(1234)
    # query encoder
    for D, dim in ((16, 10), (32, 32), (64, 50), (128, 128), (128, 96), (256, 200), (512, 512),
                   (1024, 960), (2048, 1536)):
        q = (rng.standard_normal((6, dim)) * np.array([1, 1, 100, 0.01, 1, 1])[:, None]).astype(np.float32)
        q[4] = 0.0
        q[5] = np.round(q[5] * 10)
        luts, cos, rots = [], [], []
        for i in range(6):
            lut, co, rot = r.encode_query(q[i], D)
            luts.append(lut), cos.append(co), rots.append(rot)
        out[f"E/{D}/{dim}/q"] = q
        out[f"E/{D}/{dim}/lut"] = np.stack(luts)
        out[f"E/{D}/{dim}/coeffs"] = np.stack(cos)
        out[f"E/{D}/{dim}/rot"] = np.stack(rots)

    # FastScan blocks: random valid codes + aux, reference sums and epilogues
    for D in (16, 128, 1024):
        lut, co, _ = r.encode_query(rng.standard_normal(D).astype(np.float32), D)
        for bits in (1, 2, 4):
            nblk = 8
            planes = rng.integers(0, 256, (nblk, bits, D // 8, 32), dtype=np.uint8)
            nop = rng.uniform(0, 20, (nblk, 32)).astype(np.float32)
            ipqo = rng.uniform(0.3, 1, (nblk, 32)).astype(np.float32)
            ipqo[1, :5] = 0.0  # ip_qo <= 1e-10 lanes
            ipcp = rng.uniform(-1, 1, (nblk, 32)).astype(np.float32)
            # popcounts consistent with the codes (plane-0 popcount, weighted popcount)
            pc = np.zeros((nblk, bits, 32), np.int64)
            for b in range(bits):  # byte [sp][i] holds 8 code bits of neighbour i
                pc[:, b] = PC8[planes[:, b]].sum(axis=1)
            pop = pc[:, 0].astype(np.uint16)
            wpop = sum(pc[:, b] << (bits - 1 - b) for b in range(bits)).astype(np.uint16)
            qps = np.array([[co[0], co[1], co[2], 1.0, 0.0, 0.0, 0.1],
                            [co[0], co[1], co[2], 0.93, 0.02, 0.55, -0.05]], np.float32)
            dqps = np.array([0.0, 5e-13, 37.5, 2.5e4], np.float32)
            key = f"F/{D}/b{bits}"
            out[f"{key}/lut"] = lut
            out[f"{key}/planes"] = planes
            out[f"{key}/nop"], out[f"{key}/ipqo"], out[f"{key}/ipcp"] = nop, ipqo, ipcp
            out[f"{key}/pop"], out[f"{key}/wpop"] = pop, wpop
            out[f"{key}/qps"], out[f"{key}/dqps"] = qps, dqps
            sums = np.zeros((nblk, 32), np.uint32)
            msb = np.zeros((nblk, 32), np.uint32)
            msb2 = np.zeros((nblk, 32), np.uint32)
            est = np.zeros((len(qps), len(dqps), nblk, 32), np.float32)
            lower = np.zeros_like(est)
            lower1 = np.zeros_like(est)
            for i in range(nblk):
                if bits == 1:
                    sums[i] = r.fastscan_plane(D, lut, planes[i, 0])
                    msb[i] = sums[i]
                    msb2[i] = sums[i]
                else:
                    sums[i], msb[i] = r.fastscan_nbit(D, bits, lut, planes[i])
                    msb2[i] = r.fastscan_msb(D, bits, lut, planes[i])
                for a, qp in enumerate(qps):
                    for c, dqp in enumerate(dqps):
                        if bits == 1:
                            e, lo = r.convert_1bit(D, qp, sums[i], nop[i], ipqo[i], ipcp[i], pop[i], dqp)
                            lo1 = lo
                        else:
                            lo1 = r.convert_msb(D, bits, qp, msb2[i], nop[i], ipqo[i], ipcp[i], pop[i], dqp)
                            e, lo = r.convert_nbit(D, bits, qp, sums[i], msb[i], nop[i], ipqo[i],
                                                   ipcp[i], pop[i], wpop[i], dqp)
                        est[a, c, i], lower[a, c, i], lower1[a, c, i] = e, lo, lo1
            out[f"{key}/sums"], out[f"{key}/msb"], out[f"{key}/msb2"] = sums, msb, msb2
            out[f"{key}/est"], out[f"{key}/lower"], out[f"{key}/lower1"] = est, lower, lower1
            # the same blocks with short neighbour lists: count % 8 != 0 reaches the scalar tails
            for cnt in SHORT_COUNTS:
                ec, lc, l1c = np.zeros_like(est), np.zeros_like(est), np.zeros_like(est)
                for i in range(nblk):
                    for a, qp in enumerate(qps):
                        for c, dqp in enumerate(dqps):
                            if bits == 1:
                                e, lo = r.convert_1bit(D, qp, sums[i], nop[i], ipqo[i], ipcp[i], pop[i], dqp, cnt)
                                lo1 = lo
                            else:
                                lo1 = r.convert_msb(D, bits, qp, msb2[i], nop[i], ipqo[i], ipcp[i], pop[i], dqp, cnt)
                                e, lo = r.convert_nbit(D, bits, qp, sums[i], msb[i], nop[i], ipqo[i], ipcp[i],
                                                       pop[i], wpop[i], dqp, cnt)
                            ec[a, c, i, :cnt], lc[a, c, i, :cnt], l1c[a, c, i, :cnt] = e[:cnt], lo[:cnt], lo1[:cnt]
                out[f"{key}/c{cnt}/est"], out[f"{key}/c{cnt}/lower"], out[f"{key}/c{cnt}/lower1"] = ec, lc, l1c

    for D in (16, 128, 1024):
        a = (rng.standard_normal((16, D)) * 50).astype(np.float32)
        b = rng.standard_normal((16, D)).astype(np.float32)
        out[f"X/{D}/a"], out[f"X/{D}/b"] = a, b
        out[f"X/{D}/dot"] = np.array([r.dot(a[i], b[i]) for i in range(16)], np.float32)
        out[f"X/{D}/l2"] = np.array([r.l2(a[i], b[i]) for i in range(16)], np.float32)

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
