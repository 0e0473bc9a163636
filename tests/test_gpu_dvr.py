"""Depth-Box VDB build and DVR marcher on the GPU against the CPU checker. The build is integer work:
bit-exact. The marcher's arithmetic before the final tone map is IEEE add/mul/div/sqrt/floor in the
same order on both sides, so opacity is bit-exact and colour differs by at most one 8-bit level where
powf rounds differently (tolerance: 1/255 on at most 1 % of the pixels; 1e-3 absolute otherwise)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from raytracer_amd import synthetic as syn  # noqa: E402
from raytracer_amd.data import camera as cam  # noqa: E402
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume  # noqa: E402
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable  # noqa: E402
from vol_util import HEAD_TF, SNAKE_TF, blob_volume, dvr_setup  # noqa: E402

DEV = "cuda:0"


def _build_both(oracle, vol, l2, rng, use_depth_box=True):
    ref = oracle.OracleVDB(vol, l2, 1, use_depth_box, rng)
    vdb = DepthBoxVDB(DEV)
    vdb.BuildFrom(Volume(vol, DEV), rng, apron_wid=1, log2_dims=l2, use_depth_box=use_depth_box)
    assert vdb.IsComplete()
    return vdb, ref


def _compare_build(vdb, ref):
    assert bytes(vdb.plan.param)[:124] == ref.param.tobytes()[:124]
    assert vdb.plan.leaf_num == ref.leaf_num
    np.testing.assert_array_equal(vdb.host_children, ref.child_pools)
    np.testing.assert_array_equal(vdb.host_atlas_map, ref.atlas_brick_to_node)
    np.testing.assert_array_equal(vdb.host_nodes["pos"], ref.nodes["pos"])
    n = ref.leaf_num
    np.testing.assert_array_equal(vdb.host_nodes["brickPosInAtlas"][:n], ref.nodes["brickPosInAtlas"][:n])
    np.testing.assert_array_equal(vdb.host_nodes["childPoolOffsRelative"][:-1],
                                  ref.nodes["childPoolOffsRelative"][:-1])
    np.testing.assert_array_equal(vdb.atlas_to_host(), ref.atlas)
    assert vdb.GetNodePoolsSize() == len(ref.nodes)
    assert vdb.GetChildPoolsSize() == len(ref.child_pools)
    assert vdb.GetAtlasBrickToNodeIndicesSize() == len(ref.atlas_brick_to_node)


@pytest.mark.parametrize("shape,l2,dtype,depth", [
    ((72, 80, 96), (4, 2, 2), np.uint8, True),     # edges not multiples of the brick
    ((50, 33, 70), (3, 2, 3), np.uint16, True),    # uint16 voxels, three levels
    ((64, 64, 64), (5, 4, 3), np.uint8, False),    # no depth box: apronDepWid == apronWid
    ((96, 96, 96), (5, 4, 3), np.uint8, True),
])
def test_build_matches_checker(oracle, shape, l2, dtype, depth):
    vol = blob_volume(shape, seed=5, dtype=dtype)
    rng = np.array([[0, int(0.16 * np.iinfo(dtype).max)]], dtype)
    vdb, ref = _build_both(oracle, vol, l2, rng, depth)
    _compare_build(vdb, ref)


def test_leaf_positions_made_on_the_gpu_are_the_wired_ones(oracle):
    """kk_vdb_leaf_nodes numbers the valid bricks in the pools' order on the GPU: its positions are those
    of the host-wired leaf nodes, for a brick grid with unequal edges and for one that is larger than a
    workgroup's share per thread would make trivial."""
    import ctypes as C
    from raytracer_amd import _lib
    from raytracer_amd.dpbx_vdb import NODE_DTYPE
    for shape, l2 in (((72, 80, 96), (4, 2, 2)), ((40, 200, 312), (3, 3, 3))):
        vol = blob_volume(shape, dtype=np.uint8)
        vdb, ref = _build_both(oracle, vol, l2, SNAKE_TF.empty_scalar_ranges)
        B = 1 << l2[0]
        dz, dy, dx = shape
        lx, ly, lz = -(-dx // B), -(-dy // B), -(-dz // B)
        valid = np.zeros((lz, ly, lx), np.int32)
        pos = ref.nodes["pos"][:ref.leaf_num]
        valid[pos[:, 2] // B, pos[:, 1] // B, pos[:, 0] // B] = 1
        d_valid = torch.from_numpy(valid.reshape(-1)).to(DEV)
        out = torch.zeros(ref.leaf_num * 32, dtype=torch.uint8, device=DEV)
        _lib.check(_lib.lib().kk_vdb_leaf_nodes(d_valid.data_ptr(), C.byref(vdb.plan), out.data_ptr(), None),
                   "kk_vdb_leaf_nodes")
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(NODE_DTYPE)["pos"]
        np.testing.assert_array_equal(got, pos)
        assert _lib.lib().kk_vdb_leaf_nodes(None, C.byref(vdb.plan), out.data_ptr(), None) != 0


def test_build_head_shape(oracle):
    vol = syn.head_volume()
    vdb, ref = _build_both(oracle, vol, (5, 4, 3), HEAD_TF.empty_scalar_ranges)
    _compare_build(vdb, ref)
    assert vdb.plan.param.root_lev == 1 and vdb.plan.param.vox_per_vdb == 512.0
    # the statistics the reference prints after a build (src/ray_caster/dpbx_vdb.cu:586-607)
    m = vdb.memory_statistics()
    assert m["node_pools"] == (len(ref.nodes), 32) and m["child_pools"] == (len(ref.child_pools), 4)
    assert m["atlas_brick_to_node_indices"] == (len(ref.atlas_brick_to_node), 4)
    e = int(vdb.plan.param.vox_per_atlas_brick)
    assert m["atlas_size"][:3] == tuple(int(b) * e for b in ref.param["atlasBrickPerVDB"])
    assert m["total_bytes"] == (32 * len(ref.nodes) + 4 * len(ref.child_pools) +
                                4 * m["atlas_brick_to_node_indices"][0] + m["atlas_bricks_stored"] * e ** 3)
    text = vdb.statistics_text()
    assert text.startswith("Build VDB with\n\tNode pools of size") and "Total use of memory" in text


def test_build_two_ranges(oracle):
    vol = blob_volume((64, 64, 64), seed=9)
    rng = np.array([[0, 20], [15, 45]], np.uint8)  # overlapping: pairs are counted like the reference
    vdb, ref = _build_both(oracle, vol, (3, 2, 2), rng)
    _compare_build(vdb, ref)


def test_build_rejections(capsys):
    vol = blob_volume((40, 40, 40))
    for l2 in ((0, 4, 3), (1, 1, 1), (6, 4, 3)):
        vdb = DepthBoxVDB(DEV)
        vdb.BuildFrom(Volume(vol, DEV), [[0, 10]], log2_dims=l2)
        assert not vdb.IsComplete()
    assert "DepthBoxVDB Error" in capsys.readouterr().err


def _render_both(oracle, vdb, ref, tf, dim_xyz, spaces, W, H, shading=None, orbit=0, eye_scale=1.0,
                 rate=3.0, max_alpha=0.9):
    w2s, c, proj = dvr_setup(dim_xyz, spaces, W, H, eye_scale, orbit)
    caster = RayCaster(DEV)
    caster.SetWorldToScene(w2s)
    caster.SetDepthBoxVDB(vdb)
    caster.SetTransferFunctionTexture(TransferFunctionTable(tf.dat, DEV))
    caster.SetProjection(proj)
    caster.SetCameraToWorld(c.pos, c.rotation())
    caster.SetSamplingRate(rate)
    caster.SetMaximumAlpha(max_alpha)
    if shading:
        caster.SetShading(shading["ka"], shading["kd"], shading["ks"], shading["shiness"],
                          shading["lightPosInVolumeSpace"], shading["lightCol"])
    fb = FrameBuffer(W, H, DEV)
    counters = torch.zeros(2, dtype=torch.int64, device=DEV)
    caster.RenderDepthBoxVDB(fb, counters=counters)
    torch.cuda.synchronize()
    got = fb.ReadBack()
    want, ns = oracle.dvr_render(ref, tf.dat, w2s, cam.inverse_projective(proj), c.pos, c.rotation(),
                                 W, H, max_alpha=max_alpha, sampling_rate=rate, shading=shading)
    return got, want, int(counters[0]), ns, caster, fb


def _check_images(got, want, exact_alpha=True):
    if exact_alpha:
        np.testing.assert_array_equal(got[..., 3], want[..., 3])
    d = np.abs(got[..., :3].astype(int) - want[..., :3].astype(int))
    assert d.max() <= 1, f"colour differs by {d.max()} levels"
    assert float((d.max(axis=2) > 0).mean()) <= 0.01


@pytest.mark.parametrize("orbit,eye_scale", [(0, 1.0), (137, 1.0), (0, 0.2)])
def test_render_head_shape(oracle, orbit, eye_scale):
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    got, want, ns_gpu, ns_cpu, *_ = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), 200, 150,
                                                 orbit=orbit, eye_scale=eye_scale)
    assert ns_gpu == ns_cpu and ns_cpu > 10_000
    _check_images(got, want)


def test_render_axis_aligned_rays(oracle):
    """Odd image sizes with the unrotated camera: the centre column's rays have dir.x == 0 exactly, the
    centre row's dir.y == 0, the centre pixel's both, so the DDAs run with infinite step lengths."""
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (3, 2, 2), HEAD_TF.empty_scalar_ranges)
    assert vdb.plan.param.root_lev == 2
    got, want, ns_gpu, ns_cpu, *_ = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), 201, 151)
    assert ns_gpu == ns_cpu and ns_cpu > 10_000
    _check_images(got, want)
    assert got[75, 100, 3] > 0  # the centre pixel's ray went through the head


def test_render_three_levels_uint16_and_no_depth_box(oracle):
    vol = blob_volume((50, 33, 70), seed=5, dtype=np.uint16)
    top = 65535
    # a transfer function stretched to 16 bits is still a 256-entry table over normalised scalars
    rng = np.array([[0, int(0.16 * top)]], np.uint16)
    for depth in (True, False):
        vdb, ref = _build_both(oracle, vol, (3, 2, 3), rng, depth)
        got, want, ns_gpu, ns_cpu, *_ = _render_both(oracle, vdb, ref, SNAKE_TF, (70, 33, 50), (1, 1, 1),
                                                     128, 96, orbit=61)
        assert ns_gpu == ns_cpu
        _check_images(got, want)


def test_render_with_shading(oracle):
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    shading = {"ka": 0.5, "kd": 0.5, "ks": 0.5, "shiness": 16.0,
               "lightPosInVolumeSpace": np.array([200.0, 200.0, 300.0], np.float32),
               "lightCol": np.array([1.0, 1.0, 1.0], np.float32)}
    got, want, ns_gpu, ns_cpu, *_ = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), 160, 120,
                                                 shading=shading, orbit=40)
    assert ns_gpu == ns_cpu
    # powf inside the loop: opacity stays exact (it does not depend on colour), colour within one level
    _check_images(got, want)


@pytest.mark.parametrize("dtype,depth", [(np.uint8, True), (np.uint8, False), (np.uint16, True)])
def test_kernel_for_36_voxel_bricks_equals_the_general_one(oracle, dtype, depth):
    """32-voxel leaves with the depth-box apron make 36-voxel atlas bricks, which have a kernel of their own
    (offsets of a sample's voxels made in floating point, taps of the next sample asked for before the
    present one is composited). A frame rendered with work counters goes through the general kernel, which
    is the one compared with the checker here; without counters it must be the same bytes."""
    vol = blob_volume((70, 90, 100), seed=11, dtype=dtype)
    top = 255 if dtype == np.uint8 else 65535
    rng = np.array([[0, int(0.16 * top)]], dtype)
    vdb, ref = _build_both(oracle, vol, (5, 2, 2), rng, depth)
    assert vdb.plan.param.vox_per_atlas_brick == (36 if depth else 34)
    for orbit, max_alpha in ((20, 0.9), (200, 0.5)):
        got, want, ns_gpu, ns_cpu, caster, fb = _render_both(oracle, vdb, ref, SNAKE_TF, (100, 90, 70), (1, 1, 1),
                                                             160, 120, orbit=orbit, max_alpha=max_alpha)
        assert ns_gpu == ns_cpu and ns_cpu > 10_000
        _check_images(got, want)
        fb2 = FrameBuffer(160, 120, DEV)
        caster.RenderDepthBoxVDB(fb2)
        torch.cuda.synchronize()
        assert fb2.ReadBack().tobytes() == got.tobytes()


def test_row_split_gives_the_same_frame(oracle):
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    got, want, _, _, caster, fb = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), 200, 150, orbit=15)
    fb2 = FrameBuffer(200, 150, DEV)
    for rows in ((0, 50), (50, 51), (51, 150)):
        caster.RenderDepthBoxVDB(fb2, rows=rows)
    torch.cuda.synchronize()
    assert fb2.ReadBack().tobytes() == got.tobytes()


def test_frames_in_flight_on_two_streams_equal_serial_frames(oracle):
    """Camera parameters travel by value with each launch, so a frame queued on one stream is not
    disturbed by the setters called for the next frame while it is still running."""
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    W, H = 320, 240
    _, _, _, _, caster, fb = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), W, H)
    poses = []
    for orbit in (0, 35, 70, 105, 140, 175):
        _, c, _ = dvr_setup(dim, (2, 1, 1), W, H, 1.0, orbit)
        poses.append((c.pos, c.rotation()))
    serial = []
    for pose in poses:
        caster.SetCameraToWorld(*pose)
        caster.RenderDepthBoxVDB(fb)
        torch.cuda.synchronize()
        serial.append(fb.ReadBack().copy())
    assert serial[0].tobytes() != serial[3].tobytes()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    fbs = [FrameBuffer(W, H, DEV) for _ in poses]
    for i, pose in enumerate(poses):
        caster.SetCameraToWorld(*pose)
        with torch.cuda.stream(streams[i & 1]):
            caster.RenderDepthBoxVDB(fbs[i])
    torch.cuda.synchronize()
    for i in range(len(poses)):
        assert fbs[i].ReadBack().tobytes() == serial[i].tobytes(), f"frame {i}"


def _cost_class(cost):
    """The place of the leading one and the four bits below it (kk_dvr_tile_order's sort key)."""
    c = np.minimum(cost, 65535).astype(np.int64)
    e = np.floor(np.log2(np.maximum(c, 1))).astype(np.int64)
    m = np.where(e >= 4, c >> np.maximum(e - 4, 0), c << np.maximum(4 - e, 0)) & 15
    return (e << 4) | m


def test_adaptive_tile_order_gives_the_same_frames(oracle):
    """The order in which a frame's tiles are started comes from the tile times of the frame before; the
    frames themselves are those of the plain launch order, for whole frames and for a range of rows
    whose height is no multiple of the tile, and the order the library makes is the sort of the times."""
    from raytracer_amd import _lib
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    W, H = 330, 250
    _, _, _, _, caster, fb = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), W, H)
    poses = []
    for orbit in (0, 3, 6, 90, 93, 200):
        _, c, _ = dvr_setup(dim, (2, 1, 1), W, H, 1.0, orbit)
        poses.append((c.pos, c.rotation()))
    for rows in (None, (37, 190)):
        caster.SetAdaptiveTileOrder(False)
        plain = []
        for pose in poses:
            caster.SetCameraToWorld(*pose)
            fb.rgba.zero_()
            caster.RenderDepthBoxVDB(fb, rows=rows)
            torch.cuda.synchronize()
            plain.append(fb.ReadBack().copy())
        caster.SetAdaptiveTileOrder(True)
        for i, pose in enumerate(poses):
            caster.SetCameraToWorld(*pose)
            fb.rgba.zero_()
            caster.RenderDepthBoxVDB(fb, rows=rows)
            torch.cuda.synchronize()
            assert fb.ReadBack().tobytes() == plain[i].tobytes(), f"rows {rows} frame {i}"
            st = caster._tile_state
            n_rows = H if rows is None else rows[1] - rows[0]
            assert st["n"] == -(-W // 16) * -(-n_rows // 16) == _lib.lib().kk_dvr_tile_count(W, n_rows)
            cost = st["cost"].cpu().numpy().astype(np.int64)
            order = st["order"].cpu().numpy().astype(np.int64)
            assert cost.min() >= 1 and cost.max() <= 65535
            # costly tiles first, by cost class (sixteen to a doubling), ties by tile number
            assert np.array_equal(order, np.argsort(-_cost_class(cost), kind="stable"))
    caster.SetAdaptiveTileOrder(False)


def test_adaptive_tile_order_of_compact_stripes(oracle):
    """One GPU's compact stripes (stripes 1, 4, 7, ... of the frame) with the tile order on are the
    stripes rendered in launch order, frame after frame."""
    from raytracer_amd import _lib
    dim = (64, 128, 128)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (4, 3, 3), HEAD_TF.empty_scalar_ranges)
    W, H = 330, 250
    _, _, _, _, caster, fb = _render_both(oracle, vdb, ref, HEAD_TF, dim, (2, 1, 1), W, H)
    first, stride = 1, 3
    per = int(_lib.lib().kk_stripes_per_rank(H, stride))
    plain = torch.zeros(per, 16, W, 4, dtype=torch.uint8, device=DEV)
    ordered = torch.zeros_like(plain)
    caster.SetAdaptiveTileOrder(True)
    for orbit in (0, 4, 8, 120):
        _, c, _ = dvr_setup(dim, (2, 1, 1), W, H, 1.0, orbit)
        caster.SetCameraToWorld(c.pos, c.rotation())
        caster._adaptive = False  # (the setter would drop the tile times of the frame before)
        caster.RenderDepthBoxVDB(fb, stripes=(first, stride), compact_out=plain)
        caster._adaptive = True
        caster.RenderDepthBoxVDB(fb, stripes=(first, stride), compact_out=ordered)
        torch.cuda.synchronize()
        assert torch.equal(plain, ordered), f"orbit {orbit}"
        st = caster._tile_state
        assert st["n"] == _lib.lib().kk_dvr_stripe_tile_count(W, H, first, stride) == -(-W // 16) * 5
    assert caster._tile_state["have"] and int(plain.max()) > 0
    caster.SetAdaptiveTileOrder(False)


def test_tile_order_rejections():
    from raytracer_amd import _lib
    l = _lib.lib()
    assert l.kk_dvr_tile_count(0, 16) == 0 and l.kk_dvr_tile_count(17, 1) == 2
    n = 100
    cost = torch.zeros(n, dtype=torch.int32, device=DEV)
    order = torch.zeros(n, dtype=torch.int32, device=DEV)
    ws = torch.zeros(int(l.kk_dvr_tile_order_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    assert l.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel() - 1, None) != 0
    assert l.kk_dvr_tile_order(None, n, order.data_ptr(), ws.data_ptr(), ws.numel(), None) != 0
    assert l.kk_dvr_tile_order(cost.data_ptr(), 0, order.data_ptr(), ws.data_ptr(), ws.numel(), None) != 0
    assert l.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    torch.cuda.synchronize()
    assert order.cpu().tolist() == list(range(n))  # equal costs: the tiles' own order
    cost.copy_(torch.tensor([0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 1000, 1023, 1024, 65535, 65536, 2 ** 31 - 1]
                            + [5] * (n - 16), dtype=torch.int32))
    assert l.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    torch.cuda.synchronize()
    want = np.argsort(-_cost_class(cost.cpu().numpy().astype(np.int64)), kind="stable")
    assert order.cpu().tolist() == want.tolist()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1024, 1025, 4097, 32_400, 130_000, 262_144, 262_145, 300_000])
def test_tile_order_is_the_stable_sort_by_cost_class(n):
    """Frames of up to 2^18 tiles get their order from one launch, larger ones from a pass of the radix
    sort: both are the stable sort of the tiles by cost class, costly first, whatever the workspace held
    before (most tiles of a frame share a few classes: empty space, the volume's shell)."""
    from raytracer_amd import _lib
    l = _lib.lib()
    rng = np.random.default_rng(n)
    cost = np.where(rng.random(n) < 0.7, rng.integers(1, 4, n), rng.integers(0, 70_000, n)).astype(np.int32)
    d_cost = torch.from_numpy(cost).to(DEV)
    order = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    ws = torch.full((int(l.kk_dvr_tile_order_workspace_bytes(n)),), 0xAB, dtype=torch.uint8, device=DEV)
    for _ in range(2):  # a second call finds what the first left in the workspace
        _lib.check(l.kk_dvr_tile_order(d_cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _lib.current_stream_ptr()), "kk_dvr_tile_order")
        assert l.kk_dvr_tile_order_status(ws.data_ptr(), n, _lib.current_stream_ptr()) == 0
        want = np.argsort(-_cost_class(cost.astype(np.int64)), kind="stable")
        assert np.array_equal(order.cpu().numpy().astype(np.int64), want)
        order.fill_(-1)


def test_render_without_vdb_reports_and_returns(capsys):
    RayCaster(DEV).RenderDepthBoxVDB(FrameBuffer(8, 8, DEV))
    assert "vdb or tfTex are not set" in capsys.readouterr().err


def test_full_size_head_frame_properties(oracle):
    """128 x 256 x 256 at 1080p: the frame equals the checker's on a band of rows, opacity is zero
    outside the volume's silhouette and bounded by the early-termination rule inside."""
    dim = (128, 256, 256)
    vol = syn.head_volume(dim)
    vdb, ref = _build_both(oracle, vol, (5, 4, 3), HEAD_TF.empty_scalar_ranges)
    W, H = 1920, 1080
    w2s, c, proj = dvr_setup(dim, (2, 1, 1), W, H)
    caster = RayCaster(DEV)
    caster.SetWorldToScene(w2s)
    caster.SetDepthBoxVDB(vdb)
    caster.SetTransferFunctionTexture(TransferFunctionTable(HEAD_TF.dat, DEV))
    caster.SetProjection(proj)
    caster.SetCameraToWorld(c.pos, c.rotation())
    fb = FrameBuffer(W, H, DEV)
    caster.RenderDepthBoxVDB(fb)
    torch.cuda.synchronize()
    got = fb.ReadBack()
    band = (500, 580)
    want = np.zeros((H, W, 4), np.uint8)
    oracle.dvr_render(ref, HEAD_TF.dat, w2s, cam.inverse_projective(proj), c.pos, c.rotation(), W, H,
                      rows=band, rgba=want)
    _check_images(got[band[0]:band[1]], want[band[0]:band[1]])
    assert got[0, 0, 3] == 0 and got[H // 2, W // 2, 3] > 0
    # a ray stops at the first sample that lifts opacity to maxAlpha (0.9); one more sample of at most
    # the largest table opacity can be added on top
    assert got[..., 3].max() <= 255
