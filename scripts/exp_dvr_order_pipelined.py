"""The tile order's sort off the frame's stream: frame i renders with the order made from frame i-2, and
the sort of frame i's tile times runs on a second stream beside frame i+1. fps over the orbit against the
sort on the frame's own stream (the order from frame i-1) and against the launch order.
The set-up is the first part of exp_dvr_inflight.py."""
import ctypes as C
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from raytracer_amd import _lib
if os.environ.get("KK_LIB"):  # another build of the library, for A/B runs on one box (before anything loads it)
    _lib.LIB_PATH = os.path.abspath(os.environ["KK_LIB"])
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
L = _lib.lib()
fb = FrameBuffer(W, H, dev)
n = int(L.kk_dvr_tile_count(W, H))
wsb = int(L.kk_dvr_tile_order_workspace_bytes(n))


def render(i, cost, order, stream):
    caster.SetCameraToWorld(*poses[i])
    _lib.check(L.kk_dvr_render_ordered_u8(
        C.byref(vdb.GetDeviceData()), caster.tf.tensor.data_ptr(), C.byref(caster._params),
        C.byref(caster._frame), fb.rgba.data_ptr(), W, H, 0, H, cost.data_ptr() if cost is not None else None,
        order.data_ptr() if order is not None else None, stream), "render")


def sort(cost, order, ws, stream):
    _lib.check(L.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), wsb, stream), "order")


def run(kind, frames=720):
    main = torch.cuda.Stream(dev)
    side = torch.cuda.Stream(dev)
    cost = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    order = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    ws = [torch.zeros(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]
    ev_r = [torch.cuda.Event() for _ in range(2)]
    ev_s = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    mp, sp = main.cuda_stream, side.cuda_stream

    def frame(i):
        k = i & 1
        if kind == "launch order":
            render(i, None, None, mp)
        elif kind == "same stream":
            render(i, cost[0], order[0] if i > 0 else None, mp)
            sort(cost[0], order[0], ws[0], mp)
        else:
            if i >= 2:
                main.wait_event(ev_s[k])
            render(i, cost[k], order[k] if i >= 2 else None, mp)
            ev_r[k].record(main)
            side.wait_event(ev_r[k])
            sort(cost[k], order[k], ws[k], sp)
            ev_s[k].record(side)
    for i in range(8):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 8 + frames):
        frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return f"{frames / dt:.1f} fps, {dt / frames * 1e3:.4f} ms per frame"


for rep in range(2):
    for kind in sys.argv[1:] or ("launch order", "same stream", "second stream"):
        print(kind, run(kind), flush=True)
