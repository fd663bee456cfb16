"""ctypes front-ends for the parity checkers (TEST INFRASTRUCTURE).

* ``Oracle``   -> oracle/libcph_oracle.so   (our CPU restatement; always available)
* ``RefHooks`` -> oracle/_ref/libcph_refhooks.so (the real reference kernels; only where
                  oracle/_ref was built, i.e. the authoring container or a box it travelled to)
* ``ref_module()`` -> the real reference pybind module oracle/_ref/_core*.so

Both hook libraries export the same signatures (``orc_*`` / ``ref_*``) so a test can run
one against the other.
"""
import ctypes as C
import glob
import importlib.util
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C")


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libcph_oracle.so")
    src = os.path.join(ORACLE_DIR, "cph_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


class _Hooks:
    """Common wrapper over the orc_* / ref_* kernel hooks."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.p = prefix

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def layout(self, D, bits):
        out = (C.c_long * 10)()
        rc = self._f("layout")(D, bits, out)
        assert rc == 0
        return list(out)

    def encode_query(self, q, D):
        q = _c(q, np.float32)
        lut = np.zeros((D // 4, 16), np.uint8)
        co = np.zeros(3, np.float32)
        rot = np.zeros(D, np.float32)
        f = self._f("encode_query")
        f.argtypes = [C.c_int, C.c_int, f32p, u8p, f32p, f32p]
        assert f(q.shape[0], D, q, lut, co, rot) == 0
        return lut, co, rot

    def fastscan_plane(self, D, lut, block):
        out = np.zeros(32, np.uint32)
        f = self._f("fastscan_plane")
        f.argtypes = [C.c_int, u8p, u8p, u32p]
        assert f(D, _c(lut, np.uint8), _c(block, np.uint8), out) == 0
        return out

    def fastscan_msb(self, D, bits, lut, planes):
        out = np.zeros(32, np.uint32)
        f = self._f("fastscan_msb")
        f.argtypes = [C.c_int, C.c_int, u8p, u8p, u32p]
        assert f(D, bits, _c(lut, np.uint8), _c(planes, np.uint8), out) == 0
        return out

    def fastscan_nbit(self, D, bits, lut, planes):
        o1 = np.zeros(32, np.uint32)
        o2 = np.zeros(32, np.uint32)
        f = self._f("fastscan_nbit")
        f.argtypes = [C.c_int, C.c_int, u8p, u8p, u32p, u32p]
        assert f(D, bits, _c(lut, np.uint8), _c(planes, np.uint8), o1, o2) == 0
        return o1, o2

    def convert_1bit(self, D, qp, sums, nop, ipqo, ipcp, pop, dqp, count=32):
        est = np.zeros(32, np.float32)
        lo = np.zeros(32, np.float32)
        f = self._f("convert_1bit")
        f.argtypes = [C.c_int, f32p, u32p, f32p, f32p, f32p, u16p, C.c_int, C.c_float, f32p, f32p]
        assert f(D, _c(qp, np.float32), _c(sums, np.uint32), _c(nop, np.float32),
                 _c(ipqo, np.float32), _c(ipcp, np.float32), _c(pop, np.uint16), count,
                 float(dqp), est, lo) == 0
        return est, lo

    def convert_msb(self, D, bits, qp, msb, nop, ipqo, ipcp, pop, dqp, count=32):
        lo = np.zeros(32, np.float32)
        f = self._f("convert_msb")
        f.argtypes = [C.c_int, C.c_int, f32p, u32p, f32p, f32p, f32p, u16p, C.c_int, C.c_float, f32p]
        assert f(D, bits, _c(qp, np.float32), _c(msb, np.uint32), _c(nop, np.float32),
                 _c(ipqo, np.float32), _c(ipcp, np.float32), _c(pop, np.uint16), count,
                 float(dqp), lo) == 0
        return lo

    def convert_nbit(self, D, bits, qp, nbit, msb, nop, ipqo, ipcp, pop, wpop, dqp, count=32):
        est = np.zeros(32, np.float32)
        lo = np.zeros(32, np.float32)
        f = self._f("convert_nbit")
        f.argtypes = [C.c_int, C.c_int, f32p, u32p, u32p, f32p, f32p, f32p, u16p, u16p, C.c_int,
                      C.c_float, f32p, f32p]
        assert f(D, bits, _c(qp, np.float32), _c(nbit, np.uint32), _c(msb, np.uint32),
                 _c(nop, np.float32), _c(ipqo, np.float32), _c(ipcp, np.float32),
                 _c(pop, np.uint16), _c(wpop, np.uint16), count, float(dqp), est, lo) == 0
        return est, lo

    def encode_edges(self, parent, nbrs, D, bits):
        """Data-side encoder of one vertex' edges -> (values u8[cnt,D], aux f32[cnt,3] = nop, ip_qo, ip_cp,
        pops u32[cnt,2] = msb popcount, weighted popcount)."""
        parent = _c(parent, np.float32)
        nbrs = _c(nbrs, np.float32)
        cnt, dim = nbrs.shape
        vals = np.zeros((cnt, D), np.uint8)
        aux = np.zeros((cnt, 3), np.float32)
        pops = np.zeros((cnt, 2), np.uint32)
        f = self._f("encode_edges")
        f.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, u8p, f32p, u32p]
        assert f(dim, D, bits, parent, nbrs, cnt, vals, aux, pops) == 0
        return vals, aux, pops

    def std_heap_ops(self, ops, keys, ids):
        """libstdc++ std::push_heap / std::pop_heap (comparator: greater on the key) over an operation list."""
        ops = _c(ops, np.uint8); keys = _c(keys, np.float32); ids = _c(ids, np.uint32)
        ok = np.zeros(max(1, len(keys)), np.float32)
        oi = np.zeros(max(1, len(keys)), np.uint32)
        sz = np.zeros(1, np.uint32)
        f = self._f("std_heap_ops")
        f.argtypes = [u8p, C.c_uint64, f32p, u32p, f32p, u32p, u32p]
        assert f(ops, len(ops), keys, ids, ok, oi, sz) == 0
        return ok[:sz[0]].copy(), oi[:sz[0]].copy()

    def select_neighbors(self, x, vertex, cand, R, alpha, tau, alpha_max=0.0, err=None):
        """graph/neighbor_selection.hpp:21-88 on a candidate list (ties by id); x = [n, D] padded vectors."""
        x = _c(x, np.float32)
        cand = _c(cand, np.uint32)
        out = np.zeros(32, np.uint32)
        cnt = np.zeros(1, np.uint32)
        f = self._f("select_neighbors")
        f.argtypes = [C.c_int, f32p, C.c_uint32, u32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, u32p, u32p]
        e = None if err is None else _c(err, np.float32)
        assert f(x.shape[1], x, int(vertex), cand, len(cand), int(R), float(alpha), float(tau), float(alpha_max),
                 None if e is None else e.ctypes.data, out, cnt) == 0
        return out[:cnt[0]].copy()

    def dot(self, a, b):
        out = np.zeros(1, np.float32)
        f = self._f("dot")
        f.argtypes = [C.c_int, f32p, f32p, f32p]
        assert f(len(a), _c(a, np.float32), _c(b, np.float32), out) == 0
        return out[0]

    def l2(self, a, b):
        out = np.zeros(1, np.float32)
        f = self._f("l2")
        f.argtypes = [C.c_int, f32p, f32p, f32p]
        assert f(len(a), _c(a, np.float32), _c(b, np.float32), out) == 0
        return out[0]


class Oracle(_Hooks):
    def __init__(self):
        lib = C.CDLL(build_oracle())
        super().__init__(lib, "orc_")
        lib.orc_load.restype = C.c_void_p
        lib.orc_load.argtypes = [C.c_char_p]
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_free.argtypes = [C.c_void_p]

    def rotation_signs(self, D):
        out = np.zeros((3, D), np.float32)
        self.lib.orc_rotation_signs.argtypes = [C.c_int, f32p]
        self.lib.orc_rotation_signs(D, out)
        return out

    def load(self, path):
        h = self.lib.orc_load(path.encode())
        if not h:
            raise RuntimeError(self.lib.orc_last_error().decode())
        return OracleIndex(self, h)


class OracleIndex:
    def __init__(self, orc, h):
        self.o = orc
        self.h = C.c_void_p(h)
        out = (C.c_long * 8)()
        orc.lib.orc_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        orc.lib.orc_info(self.h, out)
        (self.n, self.dim, self.D, self.bits, self.max_level, self.entry, self.vertex_bytes,
         self.n_upper) = list(out)

    def __del__(self):
        try:
            self.o.lib.orc_free(self.h)
        except Exception:
            pass

    def search_batch(self, queries, k, nthreads=0, counters=False):
        q = _c(queries, np.float32)
        n = q.shape[0]
        ids = np.zeros((n, k), np.int64)
        d = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        ctr = np.zeros((n, 9), np.uint64) if counters else None
        f = self.o.lib.orc_search_batch
        f.argtypes = [C.c_void_p, f32p, C.c_long, C.c_long, i64p, f32p,
                      np.ctypeslib.ndpointer(np.int32, flags="C"), C.c_void_p, C.c_int]
        rc = f(self.h, q, n, k, ids, d, cnt, ctr.ctypes.data if counters else None, nthreads)
        if rc != 0:
            raise RuntimeError("Search failed: invalid entry point after finalize.")
        return (ids, d, cnt, ctr) if counters else (ids, d, cnt)

    def calib_record(self, query, start):
        """One calibration sample (oracle/cph_oracle.cpp: orc_calib_record): (rec [32, 6], valid edges, dqp)."""
        rec = np.zeros((32, 6), np.float32)
        cnt = C.c_uint32()
        dqp = C.c_float()
        f = self.o.lib.orc_calib_record
        f.argtypes = [C.c_void_p, f32p, C.c_uint32, f32p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        assert f(self.h, _c(query, np.float32), int(start), rec, C.byref(cnt), C.byref(dqp)) == 0
        return rec, cnt.value, np.float32(dqp.value)

    def entry_point(self, query):
        ep = C.c_uint32()
        f = self.o.lib.orc_entry_point
        f.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_uint32)]
        f(self.h, _c(query, np.float32), C.byref(ep))
        return ep.value

    def fastscan_vertex(self, lut, qp7, vertex, dqp):
        sums = np.zeros(32, np.uint32)
        msb = np.zeros(32, np.uint32)
        est = np.zeros(32, np.float32)
        lo = np.zeros(32, np.float32)
        lo1 = np.zeros(32, np.float32)
        f = self.o.lib.orc_fastscan_vertex
        f.argtypes = [C.c_void_p, u8p, f32p, C.c_uint32, C.c_float, u32p, u32p, f32p, f32p, f32p]
        f(self.h, _c(lut, np.uint8), _c(qp7, np.float32), int(vertex), float(dqp), sums, msb,
          est, lo, lo1)
        return sums, msb, est, lo, lo1

    def neighbor_count(self, vertex):
        f = self.o.lib.orc_neighbor_count
        f.argtypes = [C.c_void_p, C.c_uint32]
        return int(f(self.h, int(vertex)))

    def exact_l2(self, query, ids):
        ids = _c(ids, np.uint32)
        out = np.zeros(len(ids), np.float32)
        f = self.o.lib.orc_exact_l2
        f.argtypes = [C.c_void_p, f32p, u32p, C.c_long, f32p]
        f(self.h, _c(query, np.float32), ids, len(ids), out)
        return out


def ref_available():
    return bool(glob.glob(os.path.join(ORACLE_DIR, "_ref", "_core*.so"))) and os.path.exists(
        os.path.join(ORACLE_DIR, "_ref", "libcph_refhooks.so"))


class RefHooks(_Hooks):
    def __init__(self):
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libcph_refhooks.so"))
        super().__init__(lib, "ref_")


_REF_MOD = None


def ref_module():
    """The compiled reference `cphnsw._core` (exposes CPIndex)."""
    global _REF_MOD
    if _REF_MOD is None:
        so = glob.glob(os.path.join(ORACLE_DIR, "_ref", "_core*.so"))[0]
        spec = importlib.util.spec_from_file_location("_core", so)
        _REF_MOD = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_REF_MOD)
    return _REF_MOD
