"""One batch of rays traced as 1, 2 or 4 chunks that alternate over two streams (each chunk its own key +
sort + trace with its own workspace; a batch starts when the one before is done on both streams): does the
next chunk's key + sort under the current chunk's trace, and its start in the current one's tail, pay inside
one batch? ms per batch of 2,097,152 rays through the 10 M-triangle soup."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
from raytracer_amd.ray_tracer import closest_hit
from raytracer_amd import _lib

dev = "cuda:0"
N, R = 10_000_000, 2_097_152
pos, tri = syn.triangle_soup(N)
bvh = LBVH(dev)
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
rays = torch.from_numpy(syn.incoherent_rays(R)).to(dev)
bvh.trace_accel()
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ref = closest_hit(bvh, rays)
torch.cuda.synchronize()


def make_out(n):
    return (torch.empty(n, dtype=torch.float32, device=dev), torch.empty((n, 2), dtype=torch.float32, device=dev),
            torch.empty(n, dtype=torch.int32, device=dev),
            torch.empty(int(_lib.lib().kk_closest_hit_accel_workspace_bytes(n)), dtype=torch.uint8, device=dev))


for chunks in (1, 2, 4, 8):
    per = R // chunks
    parts = [rays[c * per:(c + 1) * per] for c in range(chunks)]
    outs = [make_out(per) for _ in range(chunks)]
    ev = [torch.cuda.Event(), torch.cuda.Event()]

    def batch():
        for s in (0, 1):  # nothing of this batch before the last one is over on both streams
            streams[s].wait_event(ev[0]); streams[s].wait_event(ev[1])
        for c in range(chunks):
            with torch.cuda.stream(streams[c & 1]):
                closest_hit(bvh, parts[c], out=outs[c])
        for s in (0, 1):
            ev[s].record(streams[s])

    for s in (0, 1):
        ev[s].record(streams[s])
    for _ in range(3):
        batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        batch()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    same = all(torch.equal(outs[c][2], ref[2][c * per:(c + 1) * per]) and torch.equal(outs[c][0], ref[0][c * per:(c + 1) * per])
               for c in range(chunks))
    print(f"{chunks} chunk(s): {ms:.4f} ms per batch = {R / ms / 1e3:.1f} Mrays/s, results equal: {same}", flush=True)
