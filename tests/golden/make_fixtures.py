"""Generates the committed fixtures under tests/golden/ from the reference's DATA files (scene
geometry, material and camera description) and from the transfer-function tables printed in its
README. Run in the build container, where /root/reference exists:

    python tests/golden/make_fixtures.py

veach_mis.npz  : the only complete scene the reference ships (data/example-scenes-cg22/veach-mis),
                 parsed with the loader in raytracer_amd/data/mesh.py; arrays only, no source text.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracer_amd.data.mesh import OBJMesh, parse_scene_xml  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def veach_mis():
    base = os.path.join(REF, "data/example-scenes-cg22/veach-mis/veach-mis")
    scn = parse_scene_xml(base + ".xml")
    m = OBJMesh(base + ".obj", scn["lights"])
    assert m.is_complete, m.error
    G = len(m.grp_start_face_indices)
    kd = np.zeros((G, 3), np.float32)
    ks = np.zeros((G, 3), np.float32)
    tr = np.zeros((G, 3), np.float32)
    ni = np.zeros(G, np.float32)
    ns = np.zeros(G, np.float32)
    emission = np.zeros((G, 3), np.float32)
    for gi, name in m.grp2mtl_names.items():
        mt = m.name2mtls[name]
        kd[gi], ks[gi], tr[gi], ni[gi], ns[gi] = mt.kd, mt.ks, mt.tr, mt.ni, mt.ns
        if name in scn["lights"]:
            emission[gi] = scn["lights"][name][1]
    np.savez_compressed(
        os.path.join(OUT, "veach_mis.npz"),
        positions=m.positions, normals=m.normals, tex_coords=m.tex_coords,
        face_position_indices=m.face_position_indices, face_normal_indices=m.face_normal_indices,
        face_tex_coord_indices=m.face_tex_coord_indices,
        grp_start_face_indices=m.grp_start_face_indices,
        mat_kd=kd, mat_ks=ks, mat_tr=tr, mat_ni=ni, mat_ns=ns, mat_emission=emission,
        light_type=np.array([l.type for l in m.lights], np.uint8),
        light_o=np.array([l.o for l in m.lights], np.float32),
        light_r=np.array([l.r for l in m.lights], np.float32),
        light_radiance=np.array([l.radiance for l in m.lights], np.float32),
        cam_eye=scn["eye"], cam_lookat=scn["lookat"], cam_up=scn["up"],
        cam_size=np.array(scn["size"], np.int32))


if __name__ == "__main__":
    veach_mis()
    print("wrote", os.listdir(OUT))
