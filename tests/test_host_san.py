"""The host-only code of the library under sanitizers (CPU build only; GPU sanitizers are not available on this pool).

tests/host_san/host_san.cpp includes csrc/host_index.h (v2 reader / writer, repacker, query-encoder mirror),
csrc/native_file.h (GPU-native file), csrc/builder_host.h (UpperLayers, tail fit, Huber line) and
csrc/host_parallel.h, and is built twice with plain g++:
  * -fsanitize=address,undefined : good, truncated, corrupted-header, oversized-count and bit-flipped index files
    (what the reference's loader, api/hnsw_index.hpp:305-443, trusts, ours must reject with an exception); the
    builder's host statistics and the concurrent upper-layer insertion;
  * -fsanitize=thread            : UpperLayers::build on 8 threads (per-vertex spin locks), parallel_for's
    exception path, and the leader / follower policy of concurrent cph_search callers (csrc/search_coalescer.h) with a
    stand-in launch.
A sanitizer report makes the binary exit non-zero (halt_on_error / -fno-sanitize-recover)."""
import os
import shutil
import subprocess

import pytest

from golden_util import fixture_path

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_san", "host_san.cpp")
COMMON = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror"]


def _build(tmp, name, san):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not available")
    exe = os.path.join(tmp, name)
    cmd = [cxx] + COMMON + san + [SRC, "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


@pytest.fixture(scope="module")
def asan_exe(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("host_san")), "host_san_asan",
                  ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


@pytest.fixture(scope="module")
def tsan_exe(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("host_tsan")), "host_san_tsan", ["-fsanitize=thread"])


def _run(cmd, timeout=600):
    env = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1:allocator_may_return_null=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    return r.stdout


@pytest.mark.parametrize("name,bits", [("g128", 4), ("g16", 2), ("g1024", 2), ("sift96", 4)])
def test_index_files_under_asan_ubsan(asan_exe, tmp_path, name, bits):
    out = _run([asan_exe, "files", fixture_path(name, bits), str(tmp_path)])
    assert "files: ok" in out


def test_builder_host_code_under_asan_ubsan(asan_exe):
    assert "builder: ok" in _run([asan_exe, "builder"])


def test_upper_layers_and_parallel_for_under_tsan(tsan_exe):
    assert "threads: ok" in _run([tsan_exe, "threads"])


def test_search_coalescer_under_tsan(tsan_exe):
    """The leader / follower policy that gathers concurrent cph_search callers into shared launches
    (csrc/search_coalescer.h), 12 threads x 150 calls with a stand-in launch: every caller answered once with its own
    answer, errors reach every member of a failing group, groups never mix k, never more launches in flight than slots --
    and ThreadSanitizer sees no race."""
    assert "coalescer: ok" in _run([tsan_exe, "coalescer"])


def test_search_coalescer_under_asan(asan_exe):
    assert "coalescer: ok" in _run([asan_exe, "coalescer"])
