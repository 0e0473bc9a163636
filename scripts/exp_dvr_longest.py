"""The costliest tiles of a 4K frame of the marcher: where they are, what they take inside the whole frame
and what their 16-row stripe takes rendered alone. The set-up is the first part of exp_dvr_inflight.py."""
import ctypes as C
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
from raytracer_amd import _lib
L = _lib.lib()
fb = FrameBuffer(W, H, dev)
n = int(L.kk_dvr_tile_count(W, H))
tx_n = (W + 15) // 16
cost = torch.zeros(n, dtype=torch.int32, device=dev)


def render(y0, y1, cost_t, order=None):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _lib.check(L.kk_dvr_render_ordered_u8(
        C.byref(vdb.GetDeviceData()), caster.tf.tensor.data_ptr(), C.byref(caster._params),
        C.byref(caster._frame), fb.rgba.data_ptr(), W, H, y0, y1, cost_t.data_ptr(),
        order.data_ptr() if order is not None else None, _lib.current_stream_ptr()), "render")
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)


for pose_i in (8, 188, 368):
    caster.SetCameraToWorld(*poses[pose_i])
    render(0, H, cost)
    order = torch.argsort(cost, descending=True, stable=True).to(torch.int32)
    t_frame = min(render(0, H, cost, order) for _ in range(3))
    c = cost.cpu().numpy().astype("int64")
    top = c.argsort()[::-1][:4]
    q = sorted(c)
    print(f"pose {pose_i}: frame {t_frame:.3f} ms; tile times (64-cycle units) median {q[n // 2]}, 99 % {q[int(n * .99)]}, "
          f"99.9 % {q[int(n * .999)]}, max {q[-1]}; sum / 2048 / max = {c.sum() / 2048 / c.max():.3f}")
    for t in top[:2]:
        tx, ty = int(t % tx_n), int(t // tx_n)
        rows_cost = torch.zeros(tx_n, dtype=torch.int32, device=dev)
        t_alone = min(render(ty * 16, ty * 16 + 16, rows_cost) for _ in range(3))
        print(f"   tile ({tx}, {ty}) takes {c[t]} in the frame, {int(rows_cost[tx])} with its stripe rendered alone "
              f"({t_alone:.3f} ms for the stripe: {t_alone * 1e-3 / max(int(rows_cost.max()), 1) * 1e9 / 64:.2f} ns per cycle)")
