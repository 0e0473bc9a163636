"""Builds the veach-mis path-tracing set-up from the arrays of the scene the way app/rt_lbvh.cpp:163-263
does (materials per group, sphere lights, camera, 60-degree projection, identity world-to-scene)."""
import numpy as np

from .. import ray_tracer as rt
from . import camera as cam


def veach_arrays(d):
    n = d["face_position_indices"].shape[0]
    gs = d["grp_start_face_indices"]
    tris = np.zeros(n, rt.TRIANGLE_DTYPE)
    tris["normIdx"] = d["face_normal_indices"]
    tris["texCoordIdx"] = d["face_tex_coord_indices"]
    tris["grpIdx"] = np.searchsorted(gs[1:], np.arange(n), side="right")
    mats = rt.stack_structs(
        [rt.create_material(d["mat_kd"][g], d["mat_ks"][g], d["mat_tr"][g], d["mat_ni"][g],
                            d["mat_ns"][g], d["mat_emission"][g]) for g in range(len(gs))],
        rt.MATERIAL_DTYPE)
    lights = rt.stack_structs(
        [rt.create_sphere_light(d["light_r"][i], d["light_o"][i], d["light_radiance"][i])
         for i in range(len(d["light_r"]))], rt.LIGHT_DTYPE)
    return dict(triangles=tris, tri=d["face_position_indices"], positions=d["positions"],
                normals=d["normals"], lights=lights, materials=mats)


def veach_camera(d, W, H):
    c = cam.FPSCamera(d["cam_eye"], d["cam_lookat"], d["cam_up"])
    proj = cam.perspective_fov(cam.radians(60.0), W, H, 1.0, 3.0)
    return c, proj


def make_tracer(d, W, H, device="cuda:0", max_path_depth=4):
    """LBVH + RayTracer set up for the scene arrays `d` at W x H -> (tracer, arrays, camera, proj)."""
    from ..lbvh import LBVH
    arr = veach_arrays(d)
    bvh = LBVH(device)
    bvh.BuildFrom(d["positions"], d["face_position_indices"])
    assert bvh.IsComplete()
    tracer = rt.RayTracer(device)
    tracer.SetWorldToScene(np.eye(4, dtype=np.float32))
    tracer.SetMesh(d["positions"], d["normals"], d["tex_coords"], d["grp_start_face_indices"],
                   d["face_position_indices"], d["face_normal_indices"], d["face_tex_coord_indices"],
                   arr["lights"], arr["materials"])
    tracer.SetLBVH(bvh)
    tracer.SetMaxPathDepth(max_path_depth)
    c, proj = veach_camera(d, W, H)
    tracer.SetProjection(proj)
    tracer.SetCameraToWorld(c.pos, c.rotation())
    return tracer, arr, c, proj
