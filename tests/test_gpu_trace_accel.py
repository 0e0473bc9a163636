"""Structure of the trace-side 8-wide tree (kk_trace_accel_build), read back and checked against the
binary LBVH it was made from: every wide node is reachable exactly once, children are contiguous,
every triangle has exactly one record, and every slot's byte box contains the float box it stands for."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from raytracer_amd import _lib, synthetic as syn  # noqa: E402
from raytracer_amd.lbvh import LBVH  # noqa: E402

DEV = "cuda:0"
WIDE = np.dtype([("origin", "<f4", 3), ("exp", "i1", 3), ("imask", "u1"), ("childBase", "<u4"),
                 ("triBase", "<u4"), ("leafMask", "<u4"), ("counts", "<u4"), ("q", "u1", (6, 8)),
                 ("off16", "<u2", 8), ("cnt16", "<u2", 8), ("spare", "<u4", 4)])
TRI = np.dtype([("v0", "<f4", 3), ("fi", "<u4"), ("v1", "<f4", 3), ("parent", "<u4"), ("v2", "<f4", 3),
                ("k", "<u4")])
NODE = np.dtype([("lft", "<u4"), ("rht", "<u4"), ("box", "<f4", 6)])
LEAF = 0x80000000


class Info(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("header", "nodes", "tris", "parents", "total")]


def _read(bvh):
    accel = bvh.trace_accel()
    N, M = bvh._accel_sizes
    info = Info()
    _lib.check(_lib.lib().kk_trace_accel_layout(N, M, C.byref(info)), "layout")
    assert info.total == accel.numel()
    raw = accel.cpu().numpy()
    hdr = raw[info.header:info.header + 64].view(np.uint32)
    wide_num = int(hdr[10])
    assert int(hdr[6]) == N and int(hdr[7]) == max(M - 1, 1)
    wide = raw[info.nodes:info.nodes + 128 * wide_num].view(WIDE)
    tris = raw[info.tris:info.tris + 48 * N].view(TRI)
    parents = raw[info.parents:info.parents + 4 * max(M - 1, 1)].view(np.uint32)
    nodes = bvh.d_nodes[:max(M - 1, 1) * 32].cpu().numpy().view(NODE)
    return wide, tris, parents, nodes, N, M


stats = {}


def _check(bvh, pos, tri):
    wide, tris, parents, nodes, N, M = _read(bvh)
    W = len(wide)
    assert WIDE.itemsize == 128 and TRI.itemsize == 48
    imask = wide["imask"].astype(np.uint32)
    leaf_mask = wide["leafMask"] & 0xFF
    assert not (imask & leaf_mask).any()
    nchild = np.array([bin(x).count("1") for x in imask])
    # children are contiguous and every wide node but the root is the child of exactly one node
    ranges = np.zeros(W + 1, np.int64)
    has = nchild > 0
    np.add.at(ranges, wide["childBase"][has], 1)
    np.add.at(ranges, wide["childBase"][has] + nchild[has], -1)
    cover = np.cumsum(ranges)[:W]
    assert cover[0] == 0 and (cover[1:] == 1).all()
    # every triangle has one record; the records of a node are contiguous and disjoint
    assert np.array_equal(np.sort(tris["k"]), np.arange(N, dtype=np.uint32))
    faces = bvh.d_faces.cpu().numpy().view(np.uint32)
    assert np.array_equal(tris["fi"], faces[tris["k"]])
    long_run = (wide["leafMask"] & 0x100) != 0
    huge = (wide["leafMask"] & 0x200) != 0
    nib = (wide["counts"][:, None] >> (4 * np.arange(8, dtype=np.uint32))[None, :]) & 0xF   # [W, 8]
    cnt = np.where(long_run[:, None], wide["cnt16"].astype(np.int64), nib.astype(np.int64))
    is_leaf = ((leaf_mask[:, None] >> np.arange(8)) & 1) > 0
    assert ((cnt > 0) == is_leaf)[~huge].all()
    assert (nib[long_run & ~huge].max(axis=1) == 15).all() if (long_run & ~huge).any() else True
    tri_total = np.where(huge, wide["counts"], cnt.sum(axis=1)).astype(np.int64)
    off = np.cumsum(cnt, axis=1) - cnt
    assert (off[long_run & ~huge] == wide["off16"][long_run & ~huge])[is_leaf[long_run & ~huge]].all()
    tranges = np.zeros(N + 1, np.int64)
    np.add.at(tranges, wide["triBase"][tri_total > 0], 1)
    np.add.at(tranges, (wide["triBase"] + tri_total)[tri_total > 0], -1)
    assert (np.cumsum(tranges)[:N] == 1).all()
    assert int(tri_total.sum()) == N
    # a leaf of 4..120 triangles hangs under a node of its own, cut into chunks: a slot of that many
    # triangles is found only in such a node, all of whose triangles belong to one leaf
    compacted = bvh.d_compacted[:M].cpu().numpy().view(np.uint32)
    leaf_of_rec = np.searchsorted(compacted, tris["k"], side="right") - 1
    leaf_len = np.diff(np.append(compacted, np.uint32(N)).astype(np.int64))
    first_leaf = leaf_of_rec[np.minimum(wide["triBase"], N - 1)]
    last_leaf = leaf_of_rec[np.clip(wide["triBase"] + tri_total - 1, 0, N - 1)]
    run_node = ((imask == 0) & (tri_total >= 4) & (tri_total <= 120) & (first_leaf == last_leaf) &
                (leaf_len[first_leaf] == tri_total))
    mid = ((cnt >= 4) & (cnt <= 120) & is_leaf).any(axis=1) & ~huge
    assert run_node[mid].all()
    for wn in np.nonzero(run_node)[0][:2000]:
        leaves = leaf_of_rec[wide["triBase"][wn]:wide["triBase"][wn] + tri_total[wn]]
        assert (leaves == leaves[0]).all()
        assert np.count_nonzero(is_leaf[wn]) == min(8, tri_total[wn])
        assert cnt[wn][is_leaf[wn]].max() - cnt[wn][is_leaf[wn]].min() <= 1
    # ... and every leaf of that many triangles has one
    assert int(run_node.sum()) == int(((leaf_len >= 4) & (leaf_len <= 120)).sum())
    stats["run_nodes"] = int(run_node.sum())
    # vertices are the mesh's
    p = pos.reshape(-1, 3)
    t = tri.reshape(-1, 3)
    assert np.array_equal(tris["v0"], p[t[tris["fi"], 0]])
    assert np.array_equal(tris["v2"], p[t[tris["fi"], 2]])
    # the leaf parent recorded with a triangle has that triangle's leaf as a child
    compacted = bvh.d_compacted[:M].cpu().numpy().view(np.uint32)
    leaf_of_k = np.searchsorted(compacted, tris["k"], side="right") - 1
    par = nodes[tris["parent"]]
    want = (leaf_of_k.astype(np.uint32) | np.uint32(LEAF))
    assert ((par["lft"] == want) | (par["rht"] == want)).all()
    # binary parents
    if M > 1:
        inner = ~((nodes["lft"] & LEAF) != 0)
        assert (parents[nodes["lft"][inner]] == np.nonzero(inner)[0]).all()
        assert parents[0] == 0xFFFFFFFF
    # a slot's byte box contains its triangles (leaf slots): dequantise and compare
    scale = np.ldexp(np.float64(1.0), wide["exp"].astype(np.int64))            # [W, 3]
    org = wide["origin"].astype(np.float64)
    tri_min = np.minimum(np.minimum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    tri_max = np.maximum(np.maximum(tris["v0"], tris["v1"]), tris["v2"]).astype(np.float64)
    short = ~huge
    for s in range(8):
        sel = np.nonzero(((leaf_mask >> s) & 1).astype(bool) & short)[0]
        if sel.size == 0:
            continue
        lo = org[sel] + wide["q"][sel][:, 0::2, s] * scale[sel]
        hi = org[sel] + wide["q"][sel][:, 1::2, s] * scale[sel]
        first = wide["triBase"][sel] + off[sel, s]
        n_s = cnt[sel, s]
        assert (n_s > 0).all()
        for j in range(int(n_s.max())):
            m = n_s > j
            assert (lo[m] <= tri_min[first[m] + j]).all() and (hi[m] >= tri_max[first[m] + j]).all()
    return W


def test_wide_tree_of_a_soup():
    pos, tri = syn.triangle_soup(200_000)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    W = _check(bvh, pos, tri)
    M = bvh.GetDeviceData()["compactedFaceNum"]
    assert M / 8 < W < M / 2  # around four leaves per wide node


def test_wide_tree_with_long_runs_and_tiny_meshes():
    pos, tri = syn.triangle_soup(50_000, half_extent=0.02)
    pos = pos.copy().reshape(-1, 3, 3)
    pos[::7] = pos[::7] * np.float32(1e-4) + np.float32(0.5)  # thousands of triangles in one Morton cell
    pos = pos.reshape(-1, 3)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    _check(bvh, pos, tri)
    # clusters of 1..60 triangles, each inside one Morton cell: leaves of every length up to 60
    pos, tri = syn.triangle_soup(30_000, half_extent=0.02)
    sizes = 1 + (np.arange(2000) * 7919) % 60
    cluster = np.searchsorted(np.cumsum(sizes), np.arange(30_000), side="right") % 2000
    centre = syn.triangle_soup(2000, seed=5)[0].reshape(-1, 3, 3)[:, 0]
    pos = (pos.reshape(-1, 3, 3) * np.float32(1e-4) + centre[cluster][:, None, :]).astype(np.float32).reshape(-1, 3)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    _check(bvh, pos, tri)
    assert stats["run_nodes"] > 500
    # one leaf of n triangles between two far triangles that span the scene
    for n in (3, 4, 37, 120, 121):
        p, t = syn.triangle_soup(n + 2, half_extent=0.02)
        p = p.copy().reshape(-1, 3, 3)
        p[:n] = p[:n] * np.float32(1e-5) + np.float32(0.5)
        p[n] = p[n] * np.float32(1e-3)
        p[n + 1] = p[n + 1] * np.float32(1e-3) + np.float32(1.0)
        p = p.reshape(-1, 3).astype(np.float32)
        b = LBVH(DEV)
        b.BuildFrom(p, t)
        _check(b, p, t)
        assert b.GetDeviceData()["compactedFaceNum"] == 3
        assert stats["run_nodes"] == (1 if 4 <= n <= 120 else 0)
    for n in (1, 2, 3, 9):
        p, t = syn.triangle_soup(n)
        b = LBVH(DEV)
        b.BuildFrom(p, t)
        _check(b, p, t)


def test_build_without_the_lbvh_leaf_boxes():
    """kk_trace_accel_build makes the leaf boxes itself (LBVH.trace_accel hands over the ones in the LBVH
    workspace): the tree it leaves in the same buffer passes the same checks."""
    pos, tri = syn.triangle_soup(30_000, half_extent=0.03)
    pos = pos.copy().reshape(-1, 3, 3)
    pos[::5] = pos[::5] * np.float32(1e-4) + np.float32(0.25)  # leaves of several triangles too
    pos = pos.reshape(-1, 3)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    accel = bvh.trace_accel()
    N, M = bvh._accel_sizes
    accel.zero_()
    _lib.check(_lib.lib().kk_trace_accel_build(bvh.GetDeivceDataDevicePtr(), bvh.d_positions.data_ptr(),
                                               bvh.d_face_pos_idx.data_ptr(), N, M, accel.data_ptr(),
                                               accel.numel(), _lib.current_stream_ptr()),
               "kk_trace_accel_build")
    torch.cuda.synchronize()
    _check(bvh, pos, tri)
