"""Builds libcphnsw_mi355x.so (the C-ABI library, include/cphnsw_mi355x.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU; the library is built in-tree (git-ignored, but it travels
with the gpurun snapshot).
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG_DIR), "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcphnsw_mi355x.so")
SOURCES = ["cphnsw_mi355x.hip"]
HEADERS = ["cph_core.h", "builder.h", "builder_host.h", "host_parallel.h", "search_coalescer.h", "builder_pipeline.h", "device_knn.h", "device_knn_sym.h", "device_build.h", "device_buf.h", "native_file.h", "device_encode.h", "device_fastscan.h", "device_search.h", "device_stream.h", "device_heap_test.h", "host_index.h"]
# -ffp-contract=off: every fused multiply-add in the sources is explicit and placed where the
# reference has one; the compiler must not add or remove any.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 library)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(PKG_DIR)), "include", "cphnsw_mi355x.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
