"""A tile order that needs no frame before it: tiles sorted by the length of their centre ray inside the volume's
box (costly = long chord first). ms per 4K frame over part of the orbit against the launch order and the order
of the frame before; the order's own cost is not counted (torch computes the chords here).
The set-up is the first part of exp_dvr_inflight.py."""
import ctypes as C
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
import numpy as np
from raytracer_amd import _lib
L = _lib.lib()
fb = FrameBuffer(W, H, dev)
n = int(L.kk_dvr_tile_count(W, H))
TX, TY = -(-W // 16), -(-H // 16)
cost = torch.zeros(n, dtype=torch.int32, device=dev)
order = torch.zeros(n, dtype=torch.int32, device=dev)
ws = torch.zeros(int(L.kk_dvr_tile_order_workspace_bytes(n)), dtype=torch.uint8, device=dev)


def chords():
    p, f = caster._params, caster._frame
    inv_proj = torch.tensor(np.array(p.inv_proj, dtype=np.float32).reshape(4, 4).T, device=dev)
    w2s = torch.tensor(np.array(p.w2s, dtype=np.float32).reshape(4, 4).T, device=dev)
    rot = torch.tensor(np.array(f.eye_rot, dtype=np.float32).reshape(3, 3).T, device=dev)
    eye = torch.tensor(list(f.eye_pos) + [1.0], dtype=torch.float32, device=dev)
    xs = (torch.arange(TX, device=dev) * 16 + 8).float()
    ys = (torch.arange(TY, device=dev) * 16 + 8).float()
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    ndc = torch.stack([(2 * X + 1 - W) / W, (2 * Y + 1 - H) / H, torch.ones_like(X), torch.ones_like(X)], -1)
    d = (ndc @ inv_proj.T)[..., :3]
    d = torch.nn.functional.normalize(d, dim=-1) @ rot.T
    d = torch.nn.functional.normalize(d @ w2s[:3, :3].T, dim=-1)
    o = (w2s @ eye)[:3]
    inv = 1.0 / d
    t0, t1 = (0.0 - o) * inv, (1.0 - o) * inv
    tn = torch.minimum(t0, t1).amax(-1).clamp_min(0)
    tf_ = torch.maximum(t0, t1).amin(-1)
    return (tf_ - tn).clamp_min(0).reshape(-1)


# which 32^3 bricks hold a voxel the transfer function shows
_vol = syn.kingsnake_volume_torch(dim, dev)
_empty = torch.zeros(256, dtype=torch.bool, device=dev)
for lo, hi in tf.empty_scalar_ranges.tolist():
    _empty[lo:hi + 1] = True
_show = ~_empty[_vol.long().reshape(-1)].reshape(_vol.shape)
B = 32
pad = [(-s) % B for s in _show.shape]
_show = torch.nn.functional.pad(_show, (0, pad[2], 0, pad[1], 0, pad[0]))
g = [s // B for s in _show.shape]
OCC = _show.reshape(g[0], B, g[1], B, g[2], B).permute(0, 2, 4, 1, 3, 5).reshape(g[0], g[1], g[2], -1).any(-1)
print("volume tensor shape", tuple(_vol.shape), "bricks", g, "occupied", int(OCC.sum()), flush=True)
del _vol, _show
STEPS = 96


def occupancy_cost():
    p, f = caster._params, caster._frame
    inv_proj = torch.tensor(np.array(p.inv_proj, dtype=np.float32).reshape(4, 4).T, device=dev)
    w2s = torch.tensor(np.array(p.w2s, dtype=np.float32).reshape(4, 4).T, device=dev)
    rot = torch.tensor(np.array(f.eye_rot, dtype=np.float32).reshape(3, 3).T, device=dev)
    eye = torch.tensor(list(f.eye_pos) + [1.0], dtype=torch.float32, device=dev)
    xs = (torch.arange(TX, device=dev) * 16 + 8).float()
    ys = (torch.arange(TY, device=dev) * 16 + 8).float()
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    ndc = torch.stack([(2 * X + 1 - W) / W, (2 * Y + 1 - H) / H, torch.ones_like(X), torch.ones_like(X)], -1)
    d = (ndc @ inv_proj.T)[..., :3]
    d = torch.nn.functional.normalize(d, dim=-1) @ rot.T
    d = torch.nn.functional.normalize(d @ w2s[:3, :3].T, dim=-1).reshape(-1, 3)
    o = (w2s @ eye)[:3]
    inv = 1.0 / d
    t0, t1 = (0.0 - o) * inv, (1.0 - o) * inv
    tn = torch.minimum(t0, t1).amax(-1).clamp_min(0)
    tf_ = torch.maximum(t0, t1).amin(-1)
    chord = (tf_ - tn).clamp_min(0)
    k = (torch.arange(STEPS, device=dev).float() + .5) / STEPS
    pos = o + d[:, None, :] * (tn[:, None] + k[None, :] * chord[:, None])[..., None]  # unit-box coordinates (x, y, z)
    # the volume tensor is indexed [z][y][x] or [x][y][z]? dim = (dx, dy, dz); try both and keep the better
    out = []
    for perm in ((0, 1, 2), (2, 1, 0)):
        dims = torch.tensor([dim[a] for a in perm], device=dev).float()
        vox = pos[..., list(perm)] * dims if False else pos * torch.tensor(dim, device=dev).float()
        idx = (vox / B).long()
        idx = idx[..., list(perm)]
        for a in range(3):
            idx[..., a].clamp_(0, OCC.shape[a] - 1)
        hit = OCC[idx[..., 0], idx[..., 1], idx[..., 2]].float().sum(-1)
        out.append(hit * chord / STEPS)
    return out


def render(i, use_order, want_cost):
    caster.SetCameraToWorld(*poses[i])
    _lib.check(L.kk_dvr_render_ordered_u8(
        C.byref(vdb.GetDeviceData()), caster.tf.tensor.data_ptr(), C.byref(caster._params),
        C.byref(caster._frame), fb.rgba.data_ptr(), W, H, 0, H, cost.data_ptr() if want_cost else None,
        order.data_ptr() if use_order else None, _lib.current_stream_ptr()), "render")


def run(kind, frames=200):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(frames):
        caster.SetCameraToWorld(*poses[i])
        if kind.startswith("occupancy"):
            c = occupancy_cost()[int(kind[-1])]
            cost.copy_((c / c.max().clamp_min(1e-9) * 60000).to(torch.int32) + 1)
            _lib.check(L.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _lib.current_stream_ptr()), "order")
        if kind == "chord":
            c = chords()
            cost.copy_((c / c.max().clamp_min(1e-9) * 60000).to(torch.int32) + 1)
            _lib.check(L.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _lib.current_stream_ptr()), "order")
        torch.cuda.synchronize()
        ev0.record()
        render(i, kind != "launch order" and (kind != "frame before" or i > 0), kind == "frame before")
        ev1.record()
        torch.cuda.synchronize()
        tot += ev0.elapsed_time(ev1)
        if kind == "frame before":
            _lib.check(L.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _lib.current_stream_ptr()), "order")
    return round(tot / frames, 4)


for kind in ("launch order", "frame before", "chord", "occupancy 0", "occupancy 1"):
    print(kind, run(kind), "ms per frame (render kernel alone)", flush=True)
# how well the chord predicts the tile times
render(100, False, True)
torch.cuda.synchronize()
c = chords().cpu().numpy(); t = cost.cpu().numpy().astype(np.float64)
print("correlation of chord and tile time at pose 100:", round(float(np.corrcoef(c, t)[0, 1]), 3))
for j, oc in enumerate(occupancy_cost()):
    print("correlation of occupancy estimate", j, "and tile time:", round(float(np.corrcoef(oc.cpu().numpy(), t)[0, 1]), 3))
