"""Closest-hit queries and (later) the path tracer: host-side mirror of kouek::RayTracer::RayTracer
(include/ray_tracer/ray_tracer.h:23-136)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .lbvh import LBVH


def closest_hit(lbvh: LBVH, rays, count: bool = False, out=None):
    """Traces rays [R,6] (pos, dir) through a built LBVH with rayCastLBVH<false>'s result semantics
    (src/ray_tracer/ray_tracer_render.cu:192-270). Returns torch tensors (t [R], bary [R,2],
    prim [R] int32 with -1 = miss) and, when `count`, a dict of work counters."""
    l = _lib.lib()
    dev = lbvh.device
    d_rays = rays if isinstance(rays, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(rays, dtype=np.float32))
    d_rays = d_rays.to(device=dev, dtype=torch.float32).contiguous().reshape(-1, 6)
    R = d_rays.shape[0]
    if out is None:
        t = torch.empty(R, dtype=torch.float32, device=dev)
        bary = torch.empty((R, 2), dtype=torch.float32, device=dev)
        prim = torch.empty(R, dtype=torch.int32, device=dev)
        ws = torch.empty(l.kk_closest_hit_workspace_bytes(), dtype=torch.uint8, device=dev)
    else:
        t, bary, prim, ws = out
    counters = torch.zeros(4, dtype=torch.int64, device=dev) if count else None
    with torch.cuda.device(dev):
        _lib.check(l.kk_closest_hit(lbvh.GetDeivceDataDevicePtr(), lbvh.d_positions.data_ptr(),
                                    lbvh.d_face_pos_idx.data_ptr(), d_rays.data_ptr(), R,
                                    t.data_ptr(), bary.data_ptr(), prim.data_ptr(),
                                    _lib.KK_TRAVERSE_ORDERED,
                                    counters.data_ptr() if count else None, ws.data_ptr(), ws.numel(),
                                    _lib.current_stream_ptr()), "kk_closest_hit")
    if count:
        c = counters.cpu().numpy()
        return t, bary, prim, {"node_fetches": int(c[0]), "tri_tests": int(c[1]), "rays": int(c[2])}
    return t, bary, prim
