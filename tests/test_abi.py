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


def integration_md_blocks():
    """The ```python blocks of INTEGRATION.md section B (the ctypes binding a reference maintainer would add)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Bind the C-ABI"):text.index("## Several GPUs")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 3
    return blocks


def exec_integration_md(monkeypatch):
    """Executes the blocks in order in one namespace against the built library; returns the namespace."""
    from sitator_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    monkeypatch.setenv("SITATOR_LIB", _lib.LIB_PATH)
    ns = {"__name__": "integration_md"}
    for b in integration_md_blocks():
        exec(compile(b, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_md_blocks_execute_against_the_library(monkeypatch):
    """The documented binding must be executable and agree with the header: the blocks run (their own assert compares
    the struct layout with sit_abi), every struct they declare has the library's size, every symbol they bind exists and
    has the number of parameters the header declares."""
    import ctypes as C
    from sitator_amd import _lib
    ns = exec_integration_md(monkeypatch)
    lib = _lib.load()
    abi = (C.c_int32 * 6)()
    assert lib.sit_abi(abi, 6) == 6
    assert C.sizeof(ns["sit_error"]) == abi[1] == C.sizeof(_lib.SitError)
    assert C.sizeof(ns["sit_fill_params"]) == abi[2] == C.sizeof(_lib.FillParams)
    assert ns["sit_fill_params"].predict_threshold.offset == abi[3] == _lib.FillParams.predict_threshold.offset
    # field by field: the stub, the package's table and (by sit_abi) the header agree
    assert [(n, t) for n, t in ns["sit_fill_params"]._fields_] == [(n, t) for n, t in _lib.FillParams._fields_]
    assert [(n, t) for n, t in ns["sit_error"]._fields_] == [(n, t) for n, t in _lib.SitError._fields_]
    header = open(os.path.join(ROOT, "include", "sitator_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    bound = sorted(set(re.findall(r"_lib\.(sit_[a-z0-9_]+)\.argtypes", "\n".join(integration_md_blocks()))))
    assert len(bound) >= 12
    for name in bound:
        assert name in _lib.SIGNATURES, "INTEGRATION.md binds %s, which the header does not declare" % name
        fn = getattr(ns["_lib"], name)
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, header, flags=re.S)
        assert m, name
        nparams = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        assert len(fn.argtypes) == nparams == len(_lib.SIGNATURES[name][1]), "%s: %d argtypes documented, %d declared" % (
            name, len(fn.argtypes), nparams)
        for a, b in zip(fn.argtypes, _lib.SIGNATURES[name][1]):
            assert C.sizeof(a) == C.sizeof(b), "%s: an argument's width differs from the package's binding" % name


def test_header_struct_layout_matches_the_library():
    """include/sitator_hip.h compiled by the host compiler gives the sizes and offsets sit_abi reports (the header a
    maintainer reads is the header the library was built from)."""
    import ctypes as C
    import subprocess
    import tempfile
    from sitator_amd import _lib
    lib = _lib.load()
    abi = (C.c_int32 * 6)()
    lib.sit_abi(abi, 6)
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "sitator_hip.h"\n'
           'int main(void) { printf("%d %zu %zu %zu %zu\\n", SIT_ABI_VERSION, sizeof(sit_error), sizeof(sit_fill_params), '
           'offsetof(sit_fill_params, predict_threshold), offsetof(sit_error, frame)); return 0; }\n')
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "probe.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(td, "probe.c"), "-o",
                               os.path.join(td, "probe")])
        got = [int(x) for x in subprocess.check_output([os.path.join(td, "probe")]).split()]
    assert got == list(abi[:5])
    assert _lib.ABI_VERSION == abi[0]


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
