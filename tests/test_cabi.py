"""The drop-in boundary: the C-ABI library loads, exports exactly what include/*.h
declares, and the product never reaches into oracle/."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "cream_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cream_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = _header_symbols()
    assert "cream_rpe_index_fwd" in syms and "cream_rpe_index_bwd" in syms
    assert "cream_version" in syms


def test_library_exports_every_declared_symbol():
    from cream_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/cream_amd.h but not exported"


def test_ctypes_table_matches_header():
    from cream_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    _lib.load()


def test_version_handshake():
    # rpe_ops/rpe_index.py:5-8 asserts rpe_index_cpp.version() == "1.2.0"
    from cream_amd import _lib
    assert _lib.version() == "1.2.0"
    assert "gfx950" in _lib.build_info()


def test_bad_arguments_return_codes_not_crashes():
    from cream_amd import _lib
    lib = _lib.load()
    # unknown dtype, host entry (no GPU needed)
    import numpy as np
    x = np.zeros((1, 1, 2, 3), np.float32)
    idx = np.zeros((2, 2), np.int32)
    y = np.zeros((1, 1, 2, 2), np.float32)
    rc = lib.cream_rpe_index_fwd_host(y.ctypes.data, x.ctypes.data, idx.ctypes.data, 1, 1, 2, 2, 3, 99)
    assert rc == -2
    rc = lib.cream_rpe_index_fwd_host(None, x.ctypes.data, idx.ctypes.data, 1, 1, 2, 2, 3, 0)
    assert rc == -1
    rc = lib.cream_rpe_index_fwd_host(y.ctypes.data, x.ctypes.data, idx.ctypes.data, -1, 1, 2, 2, 3, 0)
    assert rc == -1
    # empty problem is a no-op success
    rc = lib.cream_rpe_index_fwd_host(None, None, None, 0, 1, 2, 2, 3, 0)
    assert rc == 0


def test_product_does_not_import_oracle():
    """Nothing under cream_amd/ may import, link or execute oracle/ (test infrastructure)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[/.]", re.M)
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cream_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                p = os.path.join(dirpath, f)
                if pat.search(open(p, errors="replace").read()):
                    bad.append(p)
    assert not bad, f"product files reference oracle/: {bad}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cream_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.CreamLibraryError):
        _lib.load()
