"""`CPIndex` — drop-in for the reference's `cphnsw.CPIndex` (src/bindings.cpp:115-240) whose
query path runs on an MI355X through the C-ABI in include/cphnsw_mi355x.h.

Same constructor, methods, argument meaning, return shapes/dtypes and exception types as the
pybind11 class.  Returned ids are the reference's internal (post-reorder) node ids.
"""
import ctypes as C

import numpy as np

from . import _lib

DEFAULT_K = 10  # constants::kDefaultK


def _as_f32(a):
    # py::array_t<float, c_style | forcecast>
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class CPIndex:
    def __init__(self, dim, bits=1, device=None):
        if dim < 0 or bits < 0:
            raise TypeError("CPIndex(): incompatible constructor arguments")  # size_t in pybind11
        if device is None:
            device = _default_device()
        self._h = C.c_void_p()
        self._dim = int(dim)
        self._bits = int(bits)
        self._device = int(device)
        _lib.check(_lib.lib().cph_create(int(dim), int(bits), int(device), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().cph_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # -- construction (host side) -------------------------------------------
    def build(self, vectors):
        v = _as_f32(vectors)
        if v.ndim != 2 or v.shape[1] != self._dim:
            raise ValueError("vectors must be a (n, dim) float32 array")
        _lib.check(_lib.lib().cph_build(self._h, v.ctypes.data, v.shape[0]))

    def finalize(self):
        _lib.check(_lib.lib().cph_finalize(self._h))

    # -- search -----------------------------------------------------------------------------
    def search(self, query, k=DEFAULT_K):
        q = _as_f32(query)
        if q.ndim != 1 or q.shape[0] != self._dim:
            raise ValueError("query must be 1D and match index dimension")
        kk = max(int(k), 1)
        ids = np.empty(kk, np.int64)
        dist = np.empty(kk, np.float32)
        m = C.c_uint64(0)
        _lib.check(_lib.lib().cph_search(self._h, q.ctypes.data, int(k), ids.ctypes.data,
                                         dist.ctypes.data, C.byref(m)))
        return ids[:m.value].copy(), dist[:m.value].copy()

    def search_batch(self, queries, k=DEFAULT_K):
        q = _as_f32(queries)
        if q.ndim != 2 or q.shape[1] != self._dim:
            raise ValueError("queries must be a (n, dim) array")
        n, k = q.shape[0], int(k)
        ids = np.empty((n, k), np.int64)
        dist = np.empty((n, k), np.float32)
        _lib.check(_lib.lib().cph_search_batch(self._h, q.ctypes.data, n, k, ids.ctypes.data,
                                               dist.ctypes.data))
        return ids, dist

    def search_batch_device(self, queries, k=DEFAULT_K, out=None, stream=None):
        """Device-resident variant: `queries` is a float32 CUDA/HIP torch tensor (n, dim) on this
        index' device; returns (ids int64, dist float32) torch tensors on the same device.  The work
        is enqueued on `stream` (default: torch's current stream) and the call does not wait for it:
        the tensors are valid in stream order.  Two batches on two streams overlap."""
        import torch
        if queries.dim() != 2 or queries.shape[1] != self._dim or queries.dtype != torch.float32:
            raise ValueError("queries must be a (n, dim) array")
        n, k = queries.shape[0], int(k)
        if not queries.is_cuda or queries.device.index != self._device:
            raise ValueError("queries must live on this index' device")
        fresh = []                       # tensors allocated here, on torch's current stream
        if not queries.is_contiguous():
            queries = queries.contiguous()
            fresh.append(queries)
        if out is None:
            ids = torch.empty((n, k), dtype=torch.int64, device=queries.device)
            dist = torch.empty((n, k), dtype=torch.float32, device=queries.device)
            fresh += [ids, dist]
        else:
            ids, dist = out
            for t, dt in ((ids, torch.int64), (dist, torch.float32)):
                if (tuple(t.shape) != (n, k) or t.dtype != dt or t.device != queries.device
                        or not t.is_contiguous()):
                    raise ValueError("out must be contiguous (n, k) int64 / float32 tensors on the queries' device")
        cur = torch.cuda.current_stream(queries.device)
        if stream is None:
            st = cur.cuda_stream
        else:
            st = getattr(stream, "cuda_stream", stream)
            if fresh and st != cur.cuda_stream:
                # The copy / the allocations above belong to the current stream: the search stream has to run after
                # them, and the caching allocator must not hand the blocks out again while the search still uses them.
                ext = stream if isinstance(stream, torch.cuda.Stream) else torch.cuda.ExternalStream(st, device=queries.device)
                ext.wait_stream(cur)
                for t in fresh:
                    t.record_stream(ext)
        _lib.check(_lib.lib().cph_search_batch_device(self._h, queries.data_ptr(), n, k, ids.data_ptr(),
                                                      dist.data_ptr(), C.c_void_p(st)))
        return ids, dist

    # -- persistence ------------------------------------------------------------------------
    def save(self, path):
        _lib.check(_lib.lib().cph_save(self._h, str(path).encode()))

    def load(self, path):
        _lib.check(_lib.lib().cph_load(self._h, str(path).encode()))

    def calib_samples_debug(self, queries, start):
        """Construction hook: the calibration sampler on given queries / start vertices: (rec [ns, 32, 6], cnt, dqp)."""
        q = _as_f32(queries)
        st = np.ascontiguousarray(start, np.uint32)
        ns = q.shape[0]
        rec = np.zeros((ns, 32, 6), np.float32)
        cnt = np.zeros(ns, np.uint32)
        dqp = np.zeros(ns, np.float32)
        _lib.check(_lib.lib().cph_calib_hook(self._h, q.ctypes.data, st.ctypes.data, ns, rec.ctypes.data, cnt.ctypes.data,
                                             dqp.ctypes.data))
        return rec, cnt, dqp

    def save_native(self, path):
        """GPU-native file (device block layout; not readable by the reference): fast to load."""
        _lib.check(_lib.lib().cph_save_native(self._h, str(path).encode()))

    def load_native(self, path):
        _lib.check(_lib.lib().cph_load_native(self._h, str(path).encode()))

    # -- properties -------------------------------------------------------------------------
    @property
    def size(self):
        n = C.c_uint64(0)
        _lib.check(_lib.lib().cph_size(self._h, C.byref(n)))
        return n.value

    @property
    def dim(self):
        return self._dim

    @property
    def is_finalized(self):
        f = C.c_int(0)
        _lib.check(_lib.lib().cph_is_finalized(self._h, C.byref(f)))
        return bool(f.value)

    # -- extras (not in the reference) ------------------------------------------------------
    def set_batch_sets(self, n_sets):
        """Batch scratch sets in rotation (1..4, default 2): batches that can be in flight together on different streams."""
        _lib.check(_lib.lib().cph_set_batch_sets(self._h, int(n_sets)))

    def set_search_params(self, slots=0, beam_capacity=0):
        _lib.check(_lib.lib().cph_set_search_params(self._h, int(slots), int(beam_capacity)))

    def last_search_stats(self):
        out = (C.c_uint64 * 12)()
        _lib.check(_lib.lib().cph_last_search_stats(self._h, out))
        keys = ("expansions", "exact_l2", "new_neighbours", "beam_pushes", "stage2_skipped",
                "rerun_queries", "kernel_us", "expansions_nothing_new", "slots", "capacity",
                "stage2_reruns", "stage2_undecided")
        return dict(zip(keys, [int(x) for x in out]))

    def synchronize(self):
        """Waits for every batch enqueued with search_batch_device."""
        _lib.check(_lib.lib().cph_synchronize(self._h))

    def last_query_expansions(self, n):
        """Vertices expanded by each of the n queries of the last batch."""
        out = np.empty(int(n), np.uint32)
        _lib.check(_lib.lib().cph_last_query_expansions(self._h, out.ctypes.data, int(n)))
        return out

    def order_queries(self, keys):
        """Launch order the search would use for these (non-negative) scheduling keys."""
        keys = np.ascontiguousarray(keys, np.float32)
        out = np.empty(len(keys), np.uint32)
        _lib.check(_lib.lib().cph_order_queries(self._h, keys.ctypes.data, len(keys), out.ctypes.data))
        return out

    def get_vectors(self, first=0, count=None):
        """Stored vectors of internal ids [first, first+count) as float32 (count, dim)."""
        if count is None:
            count = self.size - first
        out = np.empty((count, self._dim), np.float32)
        _lib.check(_lib.lib().cph_get_vectors(self._h, int(first), int(count), out.ctypes.data))
        return out

    def internal_to_input_rows(self, base, chunk=1 << 18):
        """int64[size]: input row number of every internal id, by exact row matching.
        Rows that occur several times in `base` map to one of their equal copies."""
        base = _as_f32(base)
        key = {}
        for i in range(base.shape[0] - 1, -1, -1):
            key[base[i].tobytes()] = i
        out = np.empty(self.size, np.int64)
        for lo in range(0, self.size, chunk):
            v = self.get_vectors(lo, min(chunk, self.size - lo))
            for j in range(v.shape[0]):
                out[lo + j] = key[v[j].tobytes()]
        return out

    # kernel-level hooks (parity tests)
    def encode_query(self, query):
        q = _as_f32(query)
        D = 1
        while D < self._dim:
            D *= 2
        lut = np.zeros((D // 4, 16), np.uint8)
        co = np.zeros(3, np.float32)
        _lib.check(_lib.lib().cph_encode_query(self._h, q.ctypes.data, lut.ctypes.data, co.ctypes.data))
        return lut, co

    def entry_point(self, query):
        q = _as_f32(query)
        ep = C.c_uint32(0)
        _lib.check(_lib.lib().cph_entry_point(self._h, q.ctypes.data, C.byref(ep)))
        return ep.value

    def fastscan_block(self, lut, qparams, vertex, dist_qp_sq, worst=3.402823466e+38, nn_full=False):
        lut = np.ascontiguousarray(lut, np.uint8)
        qp = np.ascontiguousarray(qparams, np.float32)
        sums = np.zeros(32, np.uint32)
        msb = np.zeros(32, np.uint32)
        est = np.zeros(32, np.float32)
        lower = np.zeros(32, np.float32)
        lower1 = np.zeros(32, np.float32)
        _lib.check(_lib.lib().cph_fastscan_block(
            self._h, lut.ctypes.data, qp.ctypes.data, int(vertex), float(dist_qp_sq), float(worst),
            int(bool(nn_full)), sums.ctypes.data, msb.ctypes.data, est.ctypes.data, lower.ctypes.data,
            lower1.ctypes.data))
        return sums, msb, est, lower, lower1

    def exact_l2(self, query, ids):
        q = _as_f32(query)
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.zeros(len(ids), np.float32)
        _lib.check(_lib.lib().cph_exact_l2(self._h, q.ctypes.data, ids.ctypes.data, len(ids),
                                           out.ctypes.data))
        return out


def knn_bruteforce(vectors, queries=None, device=None):
    """Exact 32 nearest neighbours on the GPU's matrix cores: ids uint32, squared distances float32.
    queries=None: every row of `vectors` against the others (self excluded), shape [n, 32];
    else `queries` against `vectors`, shape [nq, 32]."""
    v = _as_f32(vectors)
    n, dim = v.shape
    q = None if queries is None else _as_f32(queries)
    if q is not None and (q.ndim != 2 or q.shape[1] != dim):
        raise ValueError("queries must be a (nq, dim) array")
    rows = n if q is None else q.shape[0]
    ids = np.zeros((rows, 32), np.uint32)
    dist = np.zeros((rows, 32), np.float32)
    dev = _default_device() if device is None else int(device)
    _lib.check(_lib.lib().cph_knn_bruteforce(dev, v.ctypes.data, n, dim, None if q is None else q.ctypes.data,
                                             0 if q is None else q.shape[0], ids.ctypes.data, dist.ctypes.data))
    return ids, dist


def encode_edges(parent, nbrs, bits, device=None):
    """GPU data-side encoder of one vertex' edges (construction hook): (values u8[cnt, D], aux f32[cnt, 3] =
    nop, ip_qo, ip_cp, pops u32[cnt, 2] = msb popcount, weighted popcount)."""
    p = _as_f32(parent)
    nb = _as_f32(nbrs)
    cnt, dim = nb.shape
    D = 16
    while D < dim:
        D *= 2
    vals = np.zeros((cnt, D), np.uint8)
    aux = np.zeros((cnt, 3), np.float32)
    pops = np.zeros((cnt, 2), np.uint32)
    dev = _default_device() if device is None else int(device)
    _lib.check(_lib.lib().cph_encode_edges(dev, dim, int(bits), p.ctypes.data, nb.ctypes.data, cnt, vals.ctypes.data,
                                           aux.ctypes.data, pops.ctypes.data))
    return vals, aux, pops


def select_neighbors_debug(x, vertex, fwd, rev, R, alpha, tau, alpha_max=0.0, err=None, device=None):
    """Construction hook: the GPU selection kernel on one vertex (x = [n, D] padded vectors, fwd = 32 candidate ids with
    0xFFFFFFFF for none, rev = up to 96 more).  Returns the selected ids."""
    x = _as_f32(x)
    n, D = x.shape
    fwd = np.ascontiguousarray(fwd, np.uint32)
    rev = np.ascontiguousarray(rev, np.uint32)
    assert fwd.shape == (32,)
    out = np.zeros(32, np.uint32)
    cnt = np.zeros(1, np.uint32)
    e = None if err is None else _as_f32(err)
    dev = _default_device() if device is None else int(device)
    _lib.check(_lib.lib().cph_select_hook(dev, x.ctypes.data, n, D, int(vertex), fwd.ctypes.data, rev.ctypes.data if len(rev) else None,
                                          len(rev), int(R), float(alpha), float(tau), float(alpha_max),
                                          None if e is None else e.ctypes.data, out.ctypes.data, cnt.ctypes.data))
    return out[:cnt[0]].copy()


def heap_ops_debug(ops, keys, ids, device=None):
    """Self-test hook: runs a push (1) / pop (0) / pop-then-push (2: the way one expansion does it, with the leaf's
    ancestors fetched ahead of the pop) sequence through the beam's wave-parallel heap routines (first 255 entries in
    LDS, the rest in HBM) and returns the heap array (keys f32, ids u32)."""
    ops = np.ascontiguousarray(ops, np.uint8)
    keys = np.ascontiguousarray(keys, np.float32)
    ids = np.ascontiguousarray(ids, np.uint32)
    n_push = int((ops != 0).sum())
    assert len(keys) == n_push and len(ids) == n_push
    ok = np.zeros(max(1, n_push), np.float32)
    oi = np.zeros(max(1, n_push), np.uint32)
    sz = np.zeros(1, np.uint32)
    dev = _default_device() if device is None else int(device)
    _lib.check(_lib.lib().cph_debug_heap_ops(dev, ops.ctypes.data, len(ops), keys.ctypes.data, ids.ctypes.data, n_push,
                                             ok.ctypes.data, oi.ctypes.data, sz.ctypes.data))
    return ok[:sz[0]].copy(), oi[:sz[0]].copy()


def _default_device():
    """One process per GPU: LOCAL_RANK selects the device when launched by torch.distributed.run."""
    import os
    try:
        return int(os.environ.get("LOCAL_RANK", "0"))
    except ValueError:
        return 0


class FastScanStream:
    """Synthetic-block streaming FastScan benchmark (cph_fastscan_stream_*)."""

    def __init__(self, D, bits, n_blocks, seed=4, device=None):
        self._h = C.c_void_p()
        bb = C.c_uint64(0)
        self.D, self.bits, self.n_blocks = int(D), int(bits), int(n_blocks)
        dev = _default_device() if device is None else int(device)
        _lib.check(_lib.lib().cph_fastscan_stream_create(dev, self.D, self.bits, self.n_blocks, int(seed),
                                                         C.byref(self._h), C.byref(bb)))
        self.block_bytes = bb.value

    def run(self, reps=1):
        ms = C.c_double(0)
        ck = C.c_double(0)
        _lib.check(_lib.lib().cph_fastscan_stream_run(self._h, int(reps), C.byref(ms), C.byref(ck)))
        return ms.value, ck.value

    def export(self, first, count, ref_block_bytes):
        blocks = np.zeros(count * ref_block_bytes, np.uint8)
        lut = np.zeros((self.D // 4, 16), np.uint8)
        qp = np.zeros(7, np.float32)
        dqp = C.c_float(0)
        _lib.check(_lib.lib().cph_fastscan_stream_export(self._h, int(first), int(count), blocks.ctypes.data,
                                                         lut.ctypes.data, qp.ctypes.data, C.byref(dqp)))
        return blocks, lut, qp, dqp.value

    def eval(self, first, count):
        est = np.zeros((count, 32), np.float32)
        lower = np.zeros((count, 32), np.float32)
        _lib.check(_lib.lib().cph_fastscan_stream_eval(self._h, int(first), int(count), est.ctypes.data,
                                                       lower.ctypes.data))
        return est, lower

    def close(self):
        if self._h.value:
            _lib.lib().cph_fastscan_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
