"""CPU-only: the C-ABI library loads and exports every symbol include/sitator_hip.h declares
(no compute calls), and the ctypes table covers the header."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sitator_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sit_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sitator_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libsitator_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_missing_library_fails_loudly(monkeypatch):
    from sitator_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsitator_hip.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "sitator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                hit = re.search(r"(from|import)\s+oracle|oracle[/.]|sitator_oracle|orc_", src)
                assert hit is None, "%s reaches into the oracle: %r" % (f, hit.group(0) if hit else "")


def test_no_gpu_means_loud_failure():
    from sitator_amd import _lib
    import numpy as np
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        _lib.HipContext(np.eye(3) * 10.0)
