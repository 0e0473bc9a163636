"""CPU checker of the path-tracing frame: sanity of the restatement (the reference ships no image this
renderer produced, and its random stream - cuRAND - is unavailable: RNG stream parity unpinned)."""
import numpy as np

from raytracer_amd import ray_tracer as rt
from raytracer_amd.data import camera as cam
from scene_util import veach_arrays, veach_camera


def test_struct_layouts_match_reference_sizes():
    # sizes measured on the reference's headers: Triangle 32, Material 64, Light 80
    assert rt.TRIANGLE_DTYPE.itemsize == 32
    assert rt.MATERIAL_DTYPE.itemsize == 64 and rt.MATERIAL_DTYPE.fields["kdOverKdAddKs"][1] == 60
    assert rt.LIGHT_DTYPE.itemsize == 80 and rt.LIGHT_DTYPE.fields["area"][1] == 56


def test_lights_as_reference_builds_them(veach):
    arr = veach_arrays(veach)
    l = arr["lights"]
    # sphere radii from the OBJ light groups, area = 2 * two_pi * r^2 (material.h:89)
    np.testing.assert_allclose(l["sphere_r"], [0.049999442, 0.49999246, 1.00001252], rtol=1e-6)
    np.testing.assert_allclose(l["area"], 4 * np.pi * l["sphere_r"] ** 2, rtol=1e-6)
    m = arr["materials"]
    assert list(m["emissive"]) == [0, 0, 0, 0, 1, 1, 1, 0, 0]
    assert m["kdOverKdAddKs"][0] == np.float32(0.5) and m["kdOverKdAddKs"][7] == 1.0


def test_rng_is_uniform_and_reproducible(oracle):
    a = oracle.rng_uniforms(1234, 77, 3, 4096)
    b = oracle.rng_uniforms(1234, 77, 3, 4096)
    assert a.tobytes() == b.tobytes()
    assert a.min() > 0.0 and a.max() <= 1.0
    assert abs(a.mean() - 0.5) < 0.02
    c = oracle.rng_uniforms(1234, 78, 3, 16)
    assert (a[:16] != c).any()


def test_oracle_render_is_sane_and_split_invariant(oracle, veach):
    arr = veach_arrays(veach)
    b = oracle.lbvh_build(veach["positions"], veach["face_position_indices"])
    W, H = 96, 64
    c, proj = veach_camera(veach, W, H)
    inv = cam.inverse_projective(proj)
    eye = np.eye(4, dtype=np.float32)
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    for s in (1.0, 2.0):
        oracle.pt_render_sample(b, arr, eye, inv, c.pos, c.rotation(), acc, img, s)
    assert not np.isnan(acc).any() and 0.02 < acc.mean() < 0.6
    assert (img[..., 3] == 255).all()
    # two half-frames give the same image as one call
    acc2 = np.zeros_like(acc)
    img2 = np.zeros_like(img)
    for s in (1.0, 2.0):
        oracle.pt_render_sample(b, arr, eye, inv, c.pos, c.rotation(), acc2, img2, s, rows=(0, 20))
        oracle.pt_render_sample(b, arr, eye, inv, c.pos, c.rotation(), acc2, img2, s, rows=(20, H))
    assert acc.tobytes() == acc2.tobytes() and img.tobytes() == img2.tobytes()


def test_camera_matches_reference_glm(oracle):
    import ctypes as C
    import pytest
    ref = oracle.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for fov, w, h in [(60, 512, 512), (60, 1920, 1080), (90, 1920, 1080), (90, 3840, 2160)]:
        out = np.zeros(16, np.float32)
        ref.kkref_perspective_fov(C.c_float(fov), C.c_float(w), C.c_float(h), C.c_float(1), C.c_float(3), vp(out))
        assert np.array_equal(cam.perspective_fov(cam.radians(fov), w, h, 1, 3).reshape(-1), out)
    for eye, steps in [((0, 0, 2), 0), ((0, 0, 2), 7), ((0, 0, 2), 719), ((0, 0, 1.2880503), 360)]:
        out = np.zeros(12, np.float32)
        e = np.array(eye, np.float32)
        z = np.zeros(3, np.float32)
        u = np.array([0, 1, 0], np.float32)
        ref.kkref_camera_orbit(vp(e), vp(z), vp(u), C.c_int(steps), C.c_float(0.5), vp(out))
        k = cam.FPSCamera(e, z, u)
        for _ in range(steps):
            k.Revolve(cam._c("sqrtf", cam._dot(k.pos, k.pos)), 0.5, 0.0)
        assert np.array_equal(np.concatenate(k.GetRFUP()), out)
    for sc in ([2, 2, 2], [1024 / 795, 1024 / 795, 1]):
        out = np.zeros(16, np.float32)
        s = np.array(sc, np.float32)
        ref.kkref_w2s_from_scales(vp(s), vp(out))
        assert np.array_equal(cam.world_to_scene(s).reshape(-1), out)


def test_inspection_targets_on_cpu(oracle, veach):
    """The checker's inspection targets: the Triangles target is opaque exactly where the closest-hit
    search finds a triangle, its colour is the barycentric triple, and the AABBs target at height 0
    covers the root box."""
    from raytracer_amd.data import camera as cam
    from scene_util import veach_arrays, veach_camera
    W, H = 64, 48
    arr = veach_arrays(veach)
    c, proj = veach_camera(veach, W, H)
    b = oracle.lbvh_build(veach["positions"], veach["face_position_indices"])
    args = (b, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj), c.pos, c.rotation(), W, H)
    tri = oracle.rt_render_target(*args, "Triangles")
    box = oracle.rt_render_target(*args, "AABBs", display_tree_height=0)
    hit = tri[..., 3] == 255
    assert 0.3 < hit.mean() < 1.0
    assert (box[..., 3][hit] == 255).all()          # a hit triangle lies inside the root box
    s = tri[..., :3].astype(int).sum(axis=2)[hit]
    assert s.min() >= 252 and s.max() <= 255        # u + v + (1 - u - v) = 1 up to truncation
    pos = oracle.rt_render_target(*args, "Positions")
    assert (pos[..., 3] == tri[..., 3]).all()
    rows = oracle.rt_render_target(*args, "Normals", rows=(10, 20))
    full = oracle.rt_render_target(*args, "Normals")
    assert (rows[10:20] == full[10:20]).all() and not rows[:10].any()


def test_cpp_host_material_and_light_members_match_checker(oracle, tmp_path):
    """Material::BRDF, Light::SampleQuad and Light::SampleSphere of the C++ host headers
    (include/kouek/ray_tracer/material.h; reference include/ray_tracer/material.h:41-48, 95-114) give the
    checker's bits (both are IEEE arithmetic in glm's order over the same libm)."""
    import ctypes as C
    import os
    import subprocess
    from raytracer_amd import ray_tracer as rt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "bin", "material_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp"), "-s"])
    rng = np.random.default_rng(7)
    n = 400
    recs, want = [], []
    lib = oracle.lib()
    for i in range(n):
        kd, ks = rng.uniform(0, 1, 3).astype(np.float32), rng.uniform(0, 1, 3).astype(np.float32)
        m = rt.create_material(kd, ks, np.ones(3, np.float32), 1.5, np.float32(rng.uniform(1, 200)),
                               np.zeros(3, np.float32))
        if i % 2:
            l = rt.create_sphere_light(np.float32(rng.uniform(0.05, 2)), rng.normal(0, 3, 3).astype(np.float32),
                                       rng.uniform(1, 50, 3).astype(np.float32))
        else:
            l = rt.create_quad_light(rng.normal(0, 3, 3).astype(np.float32), rng.normal(0, 1, 3).astype(np.float32),
                                     rng.normal(0, 1, 3).astype(np.float32), rng.uniform(1, 50, 3).astype(np.float32))
        v = rng.normal(0, 1, (3, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        f = np.concatenate([v.reshape(-1), rng.uniform(0, 1, 2), rng.normal(0, 4, 3)]).astype(np.float32)
        if i % 5 == 0:
            f[6:9] = np.array([0, 1, 0], np.float32)  # |norm.y| >= .999: the other tangent branch
        mb, lb = np.asarray(m).tobytes(), np.asarray(l).tobytes()
        assert len(mb) == 64 and len(lb) == 80
        recs.append(mb + lb + f.tobytes())
        out = np.zeros(10, np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        mbuf, lbuf = np.frombuffer(mb, np.uint8).copy(), np.frombuffer(lb, np.uint8).copy()
        lib.ko_material_brdf(vp(mbuf), vp(f[0:3].copy()), vp(f[3:6].copy()), vp(f[6:9].copy()), vp(out[:3]))
        o7 = np.zeros(7, np.float32)
        lib.ko_light_sample(vp(lbuf), C.c_float(float(f[9])), C.c_float(float(f[10])), vp(f[11:14].copy()), vp(o7))
        out[3:] = o7
        want.append(out)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_bytes(np.uint32(n).tobytes() + b"".join(recs))
    subprocess.check_call([exe, str(src), str(dst)])
    got = np.frombuffer(dst.read_bytes(), np.float32).reshape(n, 10)
    want = np.stack(want)
    assert np.isfinite(want).all()
    assert got.tobytes() == want.tobytes()
