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


_NULL_ARGS_CHILD = r"""
import ctypes, sys
l = ctypes.CDLL(sys.argv[1])
for item in sys.argv[2:]:
    name, n = item.split(":")
    f = getattr(l, name)
    f.restype = ctypes.c_long
    print(name, flush=True)
    rc = f(*([ctypes.c_void_p(0)] * int(n)))
    print("=", rc, flush=True)
"""


def test_every_entry_point_refuses_null_and_zero_arguments():
    """All-zero arguments (null pointers, zero sizes) come back as an error code or a harmless size, never
    as a crash: a host that mis-wires a call must get a message, not a dead process. One child process
    makes the calls in turn and names each before it is made. (Entry points that take a struct by value,
    free or destroy a handle, or are plain setters are left out.)"""
    import subprocess
    import sys
    from raytracer_amd import _lib
    text = open(os.path.join(ROOT, "include", "kouek_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t|uint32_t)\s*(kk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    skip = {"kk_free", "kk_stream_destroy", "kk_comm_destroy", "kk_set_device", "kk_stream_synchronize",
            "kk_memset", "kk_memcpy_h2d", "kk_memcpy_d2h", "kk_malloc", "kk_stream_create"}
    items = []
    for name, params in protos:
        plist = [p.strip() for p in params.split(",")] if params.strip() not in ("", "void") else []
        by_value = any(re.search(r"\bkk_[a-z_]+\s+[a-z_]+$", p) and "*" not in p for p in plist)
        if name not in skip and not by_value:
            items.append(f"{name}:{len(plist)}")
    assert len(items) >= 60
    r = subprocess.run([sys.executable, "-c", _NULL_ARGS_CHILD, _lib.LIB_PATH] + items, capture_output=True,
                       text=True, timeout=300)
    called = [l for l in r.stdout.split("\n") if l.startswith("kk_")]
    assert r.returncode == 0, f"crashed in {called[-1] if called else '?'}: {r.stderr[-500:]}"
    assert len(called) == len(items)
    # calls that take pointers report an error (-1 = KK_ERR_INVALID as a 32-bit int, or another KK_ERR_*)
    results = dict(zip(called, [int(l[2:]) for l in r.stdout.split("\n") if l.startswith("= ")]))
    for name in ("kk_lbvh_build", "kk_closest_hit", "kk_closest_hit_accel", "kk_pt_render_sample",
                 "kk_dvr_render_u8", "kk_vdb_classify_u8", "kk_sort_pairs_u32", "kk_check_indices"):
        assert results[name] & 0xFFFFFFFF >= 0xFFFFFFF0, (name, results[name])
