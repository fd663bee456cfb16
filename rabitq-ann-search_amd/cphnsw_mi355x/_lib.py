"""ctypes binding of libcphnsw_mi355x.so — exactly the symbols include/cphnsw_mi355x.h declares."""
import ctypes as C
import importlib.util
import os
import sys

from . import build as _build

OK, INVALID_ARGUMENT, RUNTIME_ERROR, OUT_OF_MEMORY, NOT_IMPLEMENTED = range(5)

# name -> (restype, argtypes); the list is checked against the header by tests/test_abi.py
SYMBOLS = {
    "cph_last_error": (C.c_char_p, []),
    "cph_version": (C.c_int, []),
    "cph_create": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "cph_destroy": (C.c_int, [C.c_void_p]),
    "cph_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cph_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cph_save_native": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cph_load_native": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cph_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cph_dim": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cph_is_finalized": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cph_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "cph_finalize": (C.c_int, [C.c_void_p]),
    "cph_knn_bruteforce": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_void_p]),
    "cph_debug_heap_ops": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "cph_select_hook": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cph_calib_hook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cph_encode_edges": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "cph_search_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cph_search_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "cph_synchronize": (C.c_int, [C.c_void_p]),
    "cph_set_batch_sets": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cph_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                             C.POINTER(C.c_uint64)]),
    "cph_get_vectors": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "cph_set_search_params": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "cph_last_search_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cph_last_query_expansions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "cph_order_queries": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cph_encode_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cph_entry_point": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "cph_fastscan_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cph_exact_l2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cph_host_rewrite_index": (C.c_int, [C.c_char_p, C.c_char_p]),
    "cph_host_repack_block": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "cph_host_encode_query": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cph_fastscan_stream_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "cph_fastscan_stream_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cph_fastscan_stream_export": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.POINTER(C.c_float)]),
    "cph_fastscan_stream_eval": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cph_fastscan_stream_destroy": (C.c_int, [C.c_void_p]),
}

_LIB = None


def _share_hip_runtime_with_torch():
    """A PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64.  If this library pulled in the
    system copies first and torch were imported afterwards, the process would hold two HSA runtimes and torch
    would report "No HIP GPUs are available".  When torch is installed but not imported yet, load ITS HIP
    runtime first (same soname: our library then binds to it, exactly as it does when torch was imported
    first); torch itself is not imported."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads (building if needed) the HIP library.  There is no fallback: if the library cannot
    be built or loaded the product path is unavailable and this raises."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("CPH_LIB_PATH")  # diagnostic builds (e.g. -DCPH_PHASE_TIMERS)
        if not path:
            path = _build.LIB_PATH
            if _build.needs_build():
                path = _build.build_library()
        _share_hip_runtime_with_torch()
        L = C.CDLL(path)
        diagnostic = bool(os.environ.get("CPH_LIB_PATH"))
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if diagnostic:          # an A/B build of an older tree may lack the newest test hooks
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    """Maps a cph_status to the exception type pybind11 raises for the reference
    (std::invalid_argument -> ValueError, std::runtime_error -> RuntimeError, bad_alloc ->
    MemoryError)."""
    if rc == OK:
        return
    msg = lib().cph_last_error().decode("utf-8", "replace")
    if rc == INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == OUT_OF_MEMORY:
        raise MemoryError(msg)
    if rc == NOT_IMPLEMENTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
