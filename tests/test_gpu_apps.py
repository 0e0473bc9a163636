"""The two headless applications end to end on synthetic input files (app/rt_lbvh.cpp, app/dvr_dpbx_vdb.cpp)."""
import numpy as np
import pytest

from raytracer_amd.synthetic import head_volume
from raytracer_amd.data.transfer_function import README_HEAD_TF

pytestmark = pytest.mark.gpu

# the light group comes first: the reference counts the last group one face short (mesh.h:322-381)
OBJ = """mtllib s.mtl
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1
v -0.3 1.5 -0.3
v 0.3 1.5 -0.3
v 0.3 1.5 0.3
v -0.3 1.5 0.3
vn 0 1 0
vn 0 -1 0
vt 0 0
g lamp
usemtl light
f 5/1/2 6/1/2 7/1/2
f 5/1/2 7/1/2 8/1/2
g floor
usemtl grey
f 1/1/1 3/1/1 2/1/1
f 1/1/1 4/1/1 3/1/1
"""
MTL = """newmtl grey
Kd 0.6 0.6 0.6
Ks 0 0 0
Ns 1
Ni 1
newmtl light
Kd 0 0 0
Ks 0 0 0
Ns 1
Ni 1
"""
XML = """<?xml version="1.0" encoding="utf-8"?>
<camera type="perspective" width="96" height="64" fovy="60">
  <eye x="0" y="1" z="3"/>
  <lookat x="0" y="0.5" z="0"/>
  <up x="0" y="1" z="0"/>
</camera>
<light mtlname="light" type="Quad" radiance="17,12,4"/>
"""


def read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"P6\n"
        w, h = map(int, f.readline().split())
        assert f.readline() == b"255\n"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


def test_rt_lbvh_app(tmp_path, capsys):
    from raytracer_amd.app import rt_lbvh
    (tmp_path / "s.obj").write_text(OBJ)
    (tmp_path / "s.mtl").write_text(MTL)
    (tmp_path / "s.xml").write_text(XML)
    out = tmp_path / "o.ppm"
    assert rt_lbvh.main(["-obj", str(tmp_path / "s.obj"), "-spp", "8", "-o", str(out)]) == 0
    img = read_ppm(out)
    assert img.shape == (64, 96, 3)
    assert (img.max(axis=2) > 0).mean() > 0.05  # the lit floor and the lamp
    out2 = tmp_path / "n.npy"
    assert rt_lbvh.main(["-obj", str(tmp_path / "s.obj"), "-target", "Normals", "-o", str(out2)]) == 0
    n = np.load(out2)
    assert n.shape == (64, 96, 4) and (n[..., 3] == 255).any()
    assert rt_lbvh.main(["-obj", str(tmp_path / "missing.obj")]) == 1


def test_dvr_app(tmp_path):
    from raytracer_amd.app import dvr_dpbx_vdb
    dim = (32, 64, 64)
    head_volume(dim).tofile(tmp_path / "h.raw")
    (tmp_path / "h.txt").write_text(README_HEAD_TF)
    common = ["-vt", "uint8", "-vol", str(tmp_path / "h.raw"), "-dx", "32", "-dy", "64", "-dz", "64",
              "-tf", str(tmp_path / "h.txt"), "-sx", "2", "-W", "128", "-H", "96", "-ld0", "3", "-ld1", "2",
              "-ld2", "2"]
    a, b = tmp_path / "a.npy", tmp_path / "b.npy"
    assert dvr_dpbx_vdb.main(common + ["-o", str(a)]) == 0
    assert dvr_dpbx_vdb.main(common + ["-raw", "-o", str(b)]) == 0
    va, vb = np.load(a).astype(int), np.load(b).astype(int)
    assert va.shape == (96, 128, 4) and (va[..., 3] > 0).mean() > 0.05
    # the VDB marcher skips only empty space, but restarts its sample phase per brick and reads the
    # apron-padded atlas: the dense caster sees the same picture up to small differences
    assert np.abs(va - vb).mean() < 2.0 and np.percentile(np.abs(va - vb).max(axis=2), 99) <= 24
    assert dvr_dpbx_vdb.main(common[:3] + [str(tmp_path / "none.raw")] + common[4:]) == 1
