"""The C++ host classes (include/kouek/...), driven like the reference's applications drive theirs by
tests/cpp/host_api_demo.cpp, give the same tree, radiance and DVR frame as the CPU checker."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd.data import camera as cam
from scene_util import veach_arrays, veach_camera
from vol_util import HEAD_TF, dvr_setup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "host_api_demo")


def test_reference_api_compiles_with_glm(tmp_path):
    """With KOUEK_WITH_GLM the classes take the reference's own glm types (compile only; needs the
    reference checkout for its vendored glm)."""
    glm = "/root/reference/3rd"
    if not os.path.exists(os.path.join(glm, "glm", "glm.hpp")):
        pytest.skip("vendored glm not available on this machine")
    src = tmp_path / "api.cpp"
    src.write_text('#include <kouek/ray_tracer/ray_tracer.h>\n#include <kouek/ray_caster/ray_caster.h>\n'
                   'using namespace kouek;\n'
                   'void f(RayTracer::RayTracer &t, RayCaster::RayCaster &c, HIP::FrameBuffer &fb) {\n'
                   '  t.SetWorldToScene(glm::mat4(1.f)); t.SetCameraToWorld(glm::vec3(0.f), glm::mat3(1.f));\n'
                   '  t.SetProjection(glm::mat4(1.f)); t.Render(fb, glm::ivec2(4, 4));\n'
                   '  c.SetProjection(glm::mat4(1.f)); c.RenderDepthBoxVDB<uint8_t>(fb, glm::ivec2(4, 4));\n'
                   '  std::vector<glm::vec3> p; std::vector<glm::vec<3, uint32_t>> f3;\n'
                   '  RayTracer::LBVH b; b.BuildFrom({p, f3}); (void)b.GetDeivceDataDevicePtr(); }\n')
    subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-DKOUEK_WITH_GLM", f"-I{glm}",
                           f"-I{ROOT}/include", str(src)])


@pytest.mark.gpu
def test_cpp_host_classes_match_checker(oracle, veach, tmp_path):
    assert os.path.exists(BIN), "build with __graft_entry__.build() first"
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    arr = veach_arrays(veach)
    W, H, spp = 96, 64, 2
    c, proj = veach_camera(veach, W, H)
    # the scene goes in as OBJ / MTL / XML text: the program reads it with the C++ host loaders
    from test_cpp_loaders import write_scene
    write_scene(veach, str(ind))
    np.concatenate([proj.reshape(-1), c.pos, c.rotation().reshape(-1),
                    [W, H, spp]]).astype(np.float32).tofile(ind / "camera.f32")
    # volume path
    from raytracer_amd import synthetic as syn
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    l2 = (4, 3, 3)
    vol.tofile(ind / "volume.raw")
    np.array([dim[0], dim[1], dim[2], *l2], np.uint32).tofile(ind / "volume_dim.u32")
    from raytracer_amd.data.transfer_function import README_HEAD_TF
    (ind / "tf.txt").write_text(README_HEAD_TF)
    DW, DH = 120, 90
    w2s, dc, dproj = dvr_setup(dim, (2, 1, 1), DW, DH, orbit_steps=33)
    np.concatenate([w2s.reshape(-1), dproj.reshape(-1), dc.pos, dc.rotation().reshape(-1),
                    [DW, DH]]).astype(np.float32).tofile(ind / "dvr_camera.f32")

    r = subprocess.run([BIN, str(ind), str(outd)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Builded LBVH" in r.stdout and "Build VDB with" in r.stdout

    # tree
    ref = oracle.lbvh_build(veach["positions"], veach["face_position_indices"])
    hdr = np.fromfile(outd / "lbvh_hdr.u32", np.uint32)
    assert hdr.tolist() == [ref["N"], ref["M"]]
    assert np.fromfile(outd / "lbvh_nodes.bin", np.uint8).tobytes() == ref["internalNodes"].tobytes()
    log = open(outd / "lbvh.log").read()
    assert log.startswith("[OBJ]\nof:0-> v0:") and "[LBVH/Compacts]" in log and "[LBVH/INodes]\ni:0-> lc:1325I, rc:1326I" in log
    # radiance
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    for s in range(1, spp + 1):
        oracle.pt_render_sample(ref, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj),
                                c.pos, c.rotation(), acc, img, float(s))
    got = np.fromfile(outd / "pt_accum.f32", np.float32).reshape(H, W, 3)
    assert float((np.abs(got - acc).max(axis=2) > 1e-3).mean()) <= 2e-3
    assert np.fromfile(outd / "pt_spp.f32", np.float32)[0] == spp + 1
    # DVR frame
    v = oracle.OracleVDB(vol, l2, 1, True, HEAD_TF.empty_scalar_ranges)
    want, _ = oracle.dvr_render(v, HEAD_TF.dat, w2s, cam.inverse_projective(dproj), dc.pos,
                                dc.rotation(), DW, DH)
    dgot = np.fromfile(outd / "dvr_rgba.u8", np.uint8).reshape(DH, DW, 4)
    assert np.fromfile(outd / "dvr_rgba_own_stream.u8", np.uint8).tobytes() == dgot.tobytes()
    assert np.fromfile(outd / "dvr_rgba_adaptive_order.u8", np.uint8).tobytes() == dgot.tobytes()
    np.testing.assert_array_equal(dgot[..., 3], want[..., 3])
    assert np.abs(dgot[..., :3].astype(int) - want[..., :3].astype(int)).max() <= 1
    # the one-rank communicator round trip: the gathered frame is the frame; the tracer still works on
    # the tree after its broadcast (a third sample equals the checker's third)
    if os.path.exists(outd / "dvr_gathered_rgba.u8"):
        ggot = np.fromfile(outd / "dvr_gathered_rgba.u8", np.uint8)
        assert ggot.tobytes() == dgot.tobytes()
        assert np.fromfile(outd / "dvr_gathered_adaptive_rgba.u8", np.uint8).tobytes() == dgot.tobytes()
        oracle.pt_render_sample(ref, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj),
                                c.pos, c.rotation(), acc, img, float(spp + 1))
        got3 = np.fromfile(outd / "pt_accum_after_broadcast.f32", np.float32).reshape(H, W, 3)
        assert float((np.abs(got3 - acc).max(axis=2) > 1e-3).mean()) <= 2e-3
        # RayTracer::Broadcast + RenderGathered: the frame of the plain Render calls, byte for byte
        assert (np.fromfile(outd / "pt_gathered_rgba.u8", np.uint8).tobytes() ==
                np.fromfile(outd / "pt_rgba.u8", np.uint8).tobytes())
        assert float(np.fromfile(outd / "pt_gathered_spp.f32", np.float32)[0]) == spp + 1
    else:
        assert "no RCCL on this machine" in r.stderr
    # RAW-volume caster and two inspection targets through the C++ classes
    rwant, _ = oracle.raw_render(vol, HEAD_TF.dat, w2s, cam.inverse_projective(dproj), dc.pos,
                                 dc.rotation(), DW, DH)
    rgot = np.fromfile(outd / "raw_rgba.u8", np.uint8).reshape(DH, DW, 4)
    np.testing.assert_array_equal(rgot[..., 3], rwant[..., 3])
    assert np.abs(rgot[..., :3].astype(int) - rwant[..., :3].astype(int)).max() <= 1
    awant, _ = oracle.raw_render(vol, HEAD_TF.dat, w2s, cam.inverse_projective(dproj), dc.pos,
                                 dc.rotation(), DW, DH, target="AABB")
    agot = np.fromfile(outd / "raw_aabb_rgba.u8", np.uint8).reshape(DH, DW, 4)
    assert np.abs(agot.astype(int) - awant.astype(int)).max() <= 1
    for name, target, lev, layer in (("vdb_aabbs.u8", "AABBs", 0, 0),
                                     ("vdb_positions.u8", "PositionsAfterDepthSkip", 0, 1)):
        want_t = oracle.dvr_render_target(v, w2s, cam.inverse_projective(dproj), dc.pos, dc.rotation(), DW,
                                          DH, target, display_lev=lev, display_leaf_layer=layer)
        got_t = np.fromfile(outd / name, np.uint8).reshape(DH, DW, 4)
        assert np.abs(got_t.astype(int) - want_t.astype(int)).max() <= 1, name
        assert (want_t[..., 3] > 0).mean() > 0.05
    for name, target, h in (("rt_aabbs.u8", "AABBs", 5), ("rt_normals.u8", "Normals", 5)):
        want_t = oracle.rt_render_target(ref, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj),
                                         c.pos, c.rotation(), W, H, target, display_tree_height=h)
        got_t = np.fromfile(outd / name, np.uint8).reshape(H, W, 4)
        assert np.abs(got_t.astype(int) - want_t.astype(int)).max() <= 1, name
    sizes = np.fromfile(outd / "vdb_sizes.u64", np.uint64)
    assert sizes.tolist() == [len(v.nodes), len(v.child_pools), len(v.atlas_brick_to_node)]
