"""Tile orders for the marcher, made from the tile times of the frame before (kk_dvr_render_ordered_u8):
costly first (what kk_dvr_tile_order makes), costly first with the cheap ones dealt in between, and a
coarse variant. ms per 4K frame over part of the orbit, the order's own sort not counted (torch sorts here).
The set-up is the first part of exp_dvr_inflight.py."""
import ctypes as C
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
from raytracer_amd import _lib
L = _lib.lib()
fb = FrameBuffer(W, H, dev)
n = int(L.kk_dvr_tile_count(W, H))
cost = torch.zeros(n, dtype=torch.int32, device=dev)


def make_order(kind):
    srt = torch.argsort(cost, descending=True, stable=True).to(torch.int32)
    if kind == "costly first":
        return srt
    if kind == "costly and cheap alternating":
        out = torch.empty_like(srt)
        out[0::2] = srt[: (n + 1) // 2]
        out[1::2] = torch.flip(srt[(n + 1) // 2:], dims=(0,))
        return out
    if kind == "one cheap after three costly":
        k = 3 * n // 4
        out = torch.empty_like(srt)
        idx = torch.arange(n, device=dev)
        costly_pos = idx[(idx % 4) != 3][:k]
        cheap_pos = idx[(idx % 4) == 3]
        rest = n - k - cheap_pos.numel()
        out[costly_pos] = srt[:k]
        out[cheap_pos] = torch.flip(srt[k:k + cheap_pos.numel()], dims=(0,)) if rest == 0 else srt[k:k + cheap_pos.numel()]
        return out if rest == 0 else srt
    if kind.startswith("classes per doubling"):
        per = int(kind.split()[-1])
        c = cost.clamp(1, 65535).to(torch.int64)
        e = torch.floor(torch.log2(c.to(torch.float64))).to(torch.int64)
        m = torch.where(e >= 4, c >> (e - 4).clamp(min=0), c << (4 - e).clamp(min=0)) & 15
        key = (e << 4) | m
        key = key >> {16: 0, 8: 1, 4: 2, 2: 3, 1: 4}[per]
        return torch.argsort(key, descending=True, stable=True).to(torch.int32)
    if kind == "cheap first":
        return torch.flip(srt, dims=(0,))
    raise ValueError(kind)


def run(kind, frames=200):
    order = None
    def frame(i):
        nonlocal order
        caster.SetCameraToWorld(*poses[i])
        _lib.check(L.kk_dvr_render_ordered_u8(
            C.byref(vdb.GetDeviceData()), caster.tf.tensor.data_ptr(), C.byref(caster._params),
            C.byref(caster._frame), fb.rgba.data_ptr(), W, H, 0, H, cost.data_ptr(),
            order.data_ptr() if order is not None else None, _lib.current_stream_ptr()), "render")
        if kind != "launch order":
            order = make_order(kind)
    for i in range(8):
        frame(i)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(8, 8 + frames):  # the kernel alone: events around each render
        caster.SetCameraToWorld(*poses[i])
        ev0.record()
        _lib.check(L.kk_dvr_render_ordered_u8(
            C.byref(vdb.GetDeviceData()), caster.tf.tensor.data_ptr(), C.byref(caster._params),
            C.byref(caster._frame), fb.rgba.data_ptr(), W, H, 0, H, cost.data_ptr(),
            order.data_ptr() if order is not None else None, _lib.current_stream_ptr()), "render")
        ev1.record()
        torch.cuda.synchronize()
        tot += ev0.elapsed_time(ev1)
        if kind != "launch order":
            order = make_order(kind)
    return round(tot / frames, 4)


KINDS = sys.argv[1:] or ["launch order", "costly first", "classes per doubling 16", "classes per doubling 8",
                         "classes per doubling 4", "classes per doubling 2", "classes per doubling 1",
                         "costly and cheap alternating", "cheap first"]
for kind in KINDS:
    print(kind, run(kind), flush=True)
# what the tile times add up to: the frame time of a perfect schedule, were a tile's time independent of
# what runs beside it (2048 workgroups resident: 8 per CU on 256 CUs; clock64 counts at 100 MHz here or at
# the shader clock, so the ratio to the longest tile is printed as well)
c = cost.to(torch.float64)
print("tiles", n, "sum of tile times / 2048 in units of the longest tile:", round(float(c.sum() / 2048 / c.max()), 3),
      "longest tile in 64-cycle units:", int(c.max()), "mean:", round(float(c.mean()), 1))
