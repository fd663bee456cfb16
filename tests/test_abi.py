"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/cphnsw_mi355x.h declares; argument validation mirrors the
reference; no compute call is made (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cphnsw_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cph_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from cphnsw_mi355x import _lib
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"
    assert sorted(_lib.SYMBOLS) == syms, "ctypes table and header disagree"
    assert L.cph_version() >= 100


def test_argument_validation_matches_reference():
    import cphnsw_mi355x
    with pytest.raises(ValueError, match=r"Unsupported bits=3\. Supported: 1, 2, 4\."):
        cphnsw_mi355x.CPIndex(128, 3)
    with pytest.raises(ValueError, match=r"Unsupported dimension 4096 \(padded to 4096\)"):
        cphnsw_mi355x.CPIndex(4096, 1)
    with pytest.raises(ValueError):
        cphnsw_mi355x.CPIndex(0, 1)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cphnsw_mi355x
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cphnsw_mi355x.CPIndex(128, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cphnsw_mi355x.FastScanStream(128, 4, 16)


def test_product_does_not_touch_oracle():
    """The package must never import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "rabitq-ann-search_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "libcph_oracle" not in txt and "oracle_lib" not in txt and "/_ref" not in txt, f
