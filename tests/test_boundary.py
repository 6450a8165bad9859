"""CPU tests of the drop-in boundary: the C-ABI library loads and exports exactly
the symbols include/yolo2_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    from tensorflow_yolo2_amd import _lib
    return os.path.exists(_lib.LIB_PATH)


def header_symbols():
    text = open(os.path.join(ROOT, "include", "yolo2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(y2_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_tables_agree():
    from tensorflow_yolo2_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


@pytest.mark.skipif(not _built(), reason="libyolo2_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    from tensorflow_yolo2_amd import _lib
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.y2_version() >= 1


@pytest.mark.skipif(not _built(), reason="libyolo2_hip.so not built")
def test_library_exports_nothing_but_the_header():
    """the product library must not ship development entry points (y2dev_* timing
    calls that allocate their own buffers) or any other undeclared function: the exported FUNCTIONS of
    the .so are exactly the header's (kernel handle objects are data symbols, not part of the ABI)."""
    import subprocess
    from tensorflow_yolo2_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    funcs = sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in ("T", "W"))
    funcs = [f for f in funcs if not f.startswith("_Z")]          # C++ kernel stubs carry no C name
    assert funcs == header_symbols()
    assert not any("y2dev" in ln for ln in out.splitlines())


@pytest.mark.skipif(not _built(), reason="libyolo2_hip.so not built")
def test_context_rejects_widths_the_kernels_cannot_run():
    """out_chl beyond the BN kernels' row width must be an argument error, not a divide by zero"""
    import ctypes as C
    from tensorflow_yolo2_amd import _lib
    lib = _lib.load()
    spec = (C.c_int * 8)(3, 3, 32, 0, 3, 32, 4096, 0)
    h = C.c_void_p()
    assert lib.y2_ctx_create(C.byref(h), spec, 2, 2, 0, 0, 1, 8, 8, 1) < 0
    assert b"row width" in lib.y2_last_error()


@pytest.mark.skipif(not _built(), reason="libyolo2_hip.so not built")
def test_spec_and_context_planning_without_gpu():
    """context creation / planning is host-only arithmetic: the reference counts must come out."""
    import ctypes as C
    from tensorflow_yolo2_amd import _lib, engine
    lib = _lib.load()
    assert _lib.darknet19_spec(0) == engine.CORE_SPEC
    assert _lib.darknet19_spec(1, 30) == engine.CORE_SPEC + engine.det_head_spec(30)
    assert _lib.darknet19_spec(2) == engine.CORE_SPEC + engine.CLS_HEAD_SPEC
    for kind, expect in ((1, 48241690), (2, 20917112)):
        spec = _lib.darknet19_spec(kind, 30)
        flat = (C.c_int * (4 * len(spec)))(*[v for s in spec for v in s])
        h = C.c_void_p()
        tail = 1 if kind == 2 else 0
        _lib.check(lib.y2_ctx_create(C.byref(h), flat, len(spec), 18, tail, 7, 2, 224, 224, 1))
        assert lib.y2_param_count(h) == expect
        shp = (C.c_int * 4)()
        lib.y2_output_shape(h, shp)
        assert tuple(shp) == ((2, 7, 7, 30) if kind == 1 else (2, 1, 1, 1000))
        assert lib.y2_workspace_bytes(h, 1) > lib.y2_workspace_bytes(h, 0) > 0
        lib.y2_ctx_destroy(h)
    # argument errors are reported, not crashed on
    bad = (C.c_int * 4)(5, 3, 32, 0)
    h = C.c_void_p()
    assert lib.y2_ctx_create(C.byref(h), bad, 1, 1, 0, 0, 1, 8, 8, 0) < 0
    assert b"1x1 / 3x3" in lib.y2_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tensorflow_yolo2_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Y2Error):
        _lib.load()
