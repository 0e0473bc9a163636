"""Builds the veach-mis path-tracing set-up from the arrays of the scene the way app/rt_lbvh.cpp:163-263
does (materials per group, sphere lights, camera, 60-degree projection, identity world-to-scene)."""
import numpy as np

from .. import ray_tracer as rt
from . import camera as cam


def load_obj_scene(obj_path: str) -> dict:
    """OBJ + MTL + the XML next to it (camera, lights), as app/rt_lbvh.cpp:60-146, 163-214 reads them ->
    the dict of arrays veach_arrays / make_tracer take (the layout of tests/golden/veach_mis.npz)."""
    import os
    from .mesh import OBJMesh, parse_scene_xml
    base = os.path.splitext(obj_path)[0]
    scn = parse_scene_xml(base + ".xml")
    if scn is None:
        raise ValueError(f"cannot read the scene description {base}.xml")
    m = OBJMesh(obj_path, scn["lights"])
    if not m.is_complete:
        raise ValueError(m.error)
    G = len(m.grp_start_face_indices)
    kd, ks, tr = (np.zeros((G, 3), np.float32) for _ in range(3))
    ni, ns = np.zeros(G, np.float32), np.zeros(G, np.float32)
    emission = np.zeros((G, 3), np.float32)
    for gi, name in m.grp2mtl_names.items():
        if name not in m.name2mtls:  # (the reference's application throws out of .at() here)
            raise ValueError(f'group {gi} uses the material "{name}", which the material library does not define')
        mt = m.name2mtls[name]
        kd[gi], ks[gi], tr[gi], ni[gi], ns[gi] = mt.kd, mt.ks, mt.tr, mt.ni, mt.ns
        if name in scn["lights"]:
            emission[gi] = scn["lights"][name][1]
    return dict(
        positions=m.positions, normals=m.normals, tex_coords=m.tex_coords,
        face_position_indices=m.face_position_indices, face_normal_indices=m.face_normal_indices,
        face_tex_coord_indices=m.face_tex_coord_indices,
        grp_start_face_indices=m.grp_start_face_indices,
        mat_kd=kd, mat_ks=ks, mat_tr=tr, mat_ni=ni, mat_ns=ns, mat_emission=emission,
        obj_lights=m.lights, cam_eye=scn["eye"], cam_lookat=scn["lookat"], cam_up=scn["up"],
        cam_size=np.array(scn["size"], np.int32))


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
    if "obj_lights" in d:  # straight from the loader: quads and spheres
        from .mesh import QUAD
        lights = rt.stack_structs(
            [rt.create_quad_light(l.o, l.u, l.v, l.radiance) if l.type == QUAD
             else rt.create_sphere_light(l.r, l.o, l.radiance) for l in d["obj_lights"]], rt.LIGHT_DTYPE)
    else:
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
