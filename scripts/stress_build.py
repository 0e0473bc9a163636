"""The LBVH build of the 10 M-triangle soup over and over on the same buffers, alone and beside other work
on a second stream: every build must leave the same bytes (nodes, faces, run starts) and no error word."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pos, tri = syn.triangle_soup(N)
bvh = LBVH(dev)
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
torch.cuda.synchronize()


def digest():
    M = bvh.GetDeviceData()["compactedFaceNum"]
    parts = (bvh.d_nodes[:(M - 1) * 32], bvh.d_faces.view(torch.uint8), bvh.d_compacted[:M].view(torch.uint8))
    return tuple(int(p.view(torch.int32).to(torch.int64).sum()) for p in parts) + (M,)


want = digest()
side = torch.cuda.Stream(dev)
ballast = torch.rand(64 << 20, device=dev)
bad = 0
for it in range(rounds):
    if it % 2:
        with torch.cuda.stream(side):
            for _ in range(3):
                ballast.mul_(1.0000001).add_(1e-9)
    bvh.rebuild()
    torch.cuda.synchronize()
    if digest() != want:
        bad += 1
        print("build", it, "differs", flush=True)
print(f"{rounds} builds of {N} triangles, {bad} bad")
# Two different soups in turn through the same buffers: a build that read anything the build before left
# behind (in a cache of another XCD, in a counter that was not cleared) would not give the bytes of a first build.
pos_b, _ = syn.triangle_soup(N, seed=12345)
d_a = bvh.d_positions.clone()
d_b = torch.from_numpy(pos_b).to(dev).reshape(bvh.d_positions.shape)
bvh.d_positions.copy_(d_b)
bvh.rebuild()
torch.cuda.synchronize()
want_b = digest()
assert want_b != want
for it in range(rounds // 2):
    for src, w in ((d_a, want), (d_b, want_b)):
        bvh.d_positions.copy_(src)
        bvh.rebuild()
        torch.cuda.synchronize()
        if digest() != w:
            bad += 1
            print("alternating build", it, "differs", flush=True)
print(f"{rounds // 2} x 2 alternating builds, {bad} bad")
sys.exit(1 if bad else 0)
