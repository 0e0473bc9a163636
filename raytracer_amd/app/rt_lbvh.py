"""Headless app/rt_lbvh.cpp: loads an OBJ (+ MTL, + the XML of the same name), builds the LBVH, renders
`-spp` samples of the path tracer (or one frame of an inspection target) and writes the image.

    python -m raytracer_amd.app.rt_lbvh -obj scene.obj [-W 1024 -H 1024] [-spp 16] [-target Scene]
                                        [-height 0] [-depth 4] [-o out.ppm] [-log lbvh.log]
"""
import argparse
import sys
import time

import numpy as np
import torch

from ..data.scene_setup import load_obj_scene, make_tracer
from ..ray_tracer import FrameBuffer, RayTracer
from . import write_image


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="rt_lbvh", description=__doc__)
    ap.add_argument("-obj", "--obj-path", required=True, help="The path of the file of the obj.")
    ap.add_argument("-W", type=int, default=0, help="image width (default: the XML camera's)")
    ap.add_argument("-H", type=int, default=0, help="image height (default: the XML camera's)")
    ap.add_argument("-spp", type=int, default=16)
    ap.add_argument("-target", default="Scene", choices=RayTracer.RENDER_TARGETS[:-1])
    ap.add_argument("-height", type=int, default=0, help="tree height shown by the AABBs target")
    ap.add_argument("-depth", type=int, default=4, help="max depth per path")
    ap.add_argument("-o", "--out", default="rt_lbvh.ppm")
    ap.add_argument("-dev", default="cuda:0")
    args = ap.parse_args(argv)
    try:
        d = load_obj_scene(args.obj_path)
    except (OSError, ValueError) as e:
        print(e, file=sys.stderr)
        return 1
    W = args.W or int(d["cam_size"][0])
    H = args.H or int(d["cam_size"][1])
    t0 = time.perf_counter()
    tracer, arr, c, proj = make_tracer(d, W, H, args.dev, max_path_depth=args.depth)
    torch.cuda.synchronize()
    print(f"LBVH of {len(d['face_position_indices'])} triangles built in "
          f"{(time.perf_counter() - t0) * 1e3:.1f} ms (incl. upload)")
    tracer.SetDisplayTreeHeight(args.height)
    fb = FrameBuffer(W, H, args.dev)
    t0 = time.perf_counter()
    n = args.spp if args.target == "Scene" else 1
    for _ in range(n):
        tracer.Render(fb, target=args.target)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.target}: {n} pass(es) of {W}x{H} in {dt * 1e3:.1f} ms")
    write_image(args.out, fb.ReadBack())
    print("wrote", args.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
