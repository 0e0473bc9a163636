"""The C-ABI library loads and exports every symbol the public header declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(kk_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_header_symbols():
    from raytracer_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build() first"
    l = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 15
    missing = [n for n in sorted(declared) if not hasattr(l, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_build_info_and_error_string():
    from raytracer_amd import _lib
    l = _lib.lib()
    assert l.kk_build_info().decode().startswith("gfx950")
    lay = _lib.LbvhWsLayout()
    assert l.kk_lbvh_workspace_layout(0, 0, 0, ctypes.byref(lay)) == -1  # face_num == 0 rejected
    assert b"face_num" in l.kk_last_error()
    assert l.kk_lbvh_workspace_layout(1000, 3000, 0, ctypes.byref(lay)) == 0
    assert lay.total_bytes > 1000 * 40 and lay.unsorted_mortons == 0
    assert l.kk_lbvh_workspace_layout(1000, 3000, 1, ctypes.byref(lay)) == 0
    assert lay.unsorted_mortons != 0


def test_missing_library_fails_loudly(monkeypatch):
    from raytracer_amd import _lib
    import pytest
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libkouek_hip.so")
    with pytest.raises(_lib.KoukeError):
        _lib.lib()


def test_product_package_never_imports_the_checker():
    """oracle/ is test infrastructure: nothing under raytracer_amd/ may import, load or run it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                if re.search(r"\boracle\b|kkoracle|libkkref", text):
                    offenders.append(os.path.relpath(os.path.join(dirpath, fn), ROOT))
    assert not offenders, offenders
