"""Writes tests/golden/oracle_pins.json: SHA-1 of what the CPU checker (oracle/) produces on small fixed
inputs - tree, hits, a path-traced frame, a Depth-Box VDB and a marched frame. The GPU tests compare the HIP
paths with the checker; these pins hold the checker itself still, so that an edit under oracle/ that changes
its results is seen even where the HIP code would change with it. They were made by this checker (they
are no reference outputs) and are regenerated only on purpose:  python tests/golden/make_oracle_pins.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(*arrays) -> str:
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def compute() -> dict:
    from oracle import kkoracle as ko
    from raytracer_amd import synthetic as syn
    from raytracer_amd.data import camera as cam
    from raytracer_amd.data.scene_setup import veach_arrays, veach_camera
    from vol_util import HEAD_TF, blob_volume, dvr_setup
    pins = {}
    pos, tri = syn.triangle_soup(5000, seed=1)
    b = ko.lbvh_build(pos, tri, refit_mode=1)
    pins["lbvh_soup_5000"] = sha(b["faces"], b["compactedFaces"], b["internalNodes"], b["leafAABBs"],
                                 b["uniqueMortons"], b["rootAABB"])
    rays = syn.incoherent_rays(512, seed=2)
    t, bary, prim, _ = ko.closest_hit(b, pos, tri, rays)
    pins["closest_hit_512"] = sha(t, bary, prim)
    v = dict(np.load(os.path.join(ROOT, "tests", "golden", "veach_mis.npz")))
    vb = ko.lbvh_build(v["positions"], v["face_position_indices"])
    pins["lbvh_veach_mis"] = sha(vb["faces"], vb["compactedFaces"], vb["internalNodes"])
    arr = veach_arrays(v)
    W, H = 64, 48
    c, proj = veach_camera(v, W, H)
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    for s in (1.0, 2.0):
        ko.pt_render_sample(vb, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj), c.pos, c.rotation(),
                            acc, img, s)
    pins["path_trace_64x48_2spp_rgba8"] = sha(img)
    pins["rng_uniforms"] = sha(ko.rng_uniforms(1234, 77, 3, 64))
    vol = blob_volume((40, 48, 56), seed=3)
    rng = np.array([[0, 40]], np.uint8)
    vdb = ko.OracleVDB(vol, (3, 2, 2), 1, True, rng)
    pins["vdb_blob_40x48x56"] = sha(vdb.param.tobytes()[:124] if hasattr(vdb.param, "tobytes") else bytes(vdb.param)[:124],
                                    vdb.child_pools, vdb.atlas_brick_to_node, vdb.nodes["pos"], vdb.atlas)
    w2s, c2, proj2 = dvr_setup((56, 48, 40), (1, 1, 1), 64, 48, 1.0, 17)
    frame, ns = ko.dvr_render(vdb, HEAD_TF.dat, w2s, cam.inverse_projective(proj2), c2.pos, c2.rotation(), 64, 48,
                              max_alpha=0.9, sampling_rate=3.0, shading=None)
    pins["dvr_64x48_alpha"] = sha(frame[..., 3])
    pins["dvr_64x48_samples"] = int(ns)
    return pins


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "oracle_pins.json")
    json.dump(compute(), open(out, "w"), indent=1)
    print(open(out).read())
