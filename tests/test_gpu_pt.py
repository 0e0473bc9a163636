"""Path-tracing frame on the GPU against the CPU checker: same generator values, radiance within 1e-3
absolute (the two sides differ only in libm vs device transcendental rounding, which can flip a rare
shadow or lobe decision)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from raytracer_amd import _lib  # noqa: E402
from raytracer_amd.data import camera as cam  # noqa: E402
from raytracer_amd.lbvh import LBVH  # noqa: E402
from raytracer_amd.ray_tracer import FrameBuffer, RayTracer  # noqa: E402
from scene_util import veach_arrays, veach_camera  # noqa: E402

DEV = "cuda:0"
TOL = 1e-3


def test_generator_matches_checker(oracle):
    out = torch.zeros(64, dtype=torch.float32, device=DEV)
    for seed, pixel, sample in [(1234, 0, 1), (1234, 123456, 7), (99, 2 ** 31 + 5, 1024)]:
        _lib.check(_lib.lib().kk_rng_uniforms(seed, pixel, sample, 64, out.data_ptr(), None), "rng")
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == oracle.rng_uniforms(seed, pixel, sample, 64).tobytes()


def _setup(veach, W, H):
    arr = veach_arrays(veach)
    bvh = LBVH(DEV)
    bvh.BuildFrom(veach["positions"], veach["face_position_indices"])
    assert bvh.IsComplete()
    tracer = RayTracer(DEV)
    tracer.SetWorldToScene(np.eye(4, dtype=np.float32))
    tracer.SetMesh(veach["positions"], veach["normals"], veach["tex_coords"],
                   veach["grp_start_face_indices"], veach["face_position_indices"],
                   veach["face_normal_indices"], veach["face_tex_coord_indices"], arr["lights"],
                   arr["materials"])
    tracer.SetLBVH(bvh)
    tracer.SetMaxPathDepth(4)
    c, proj = veach_camera(veach, W, H)
    tracer.SetProjection(proj)
    tracer.SetCameraToWorld(c.pos, c.rotation())
    return tracer, arr, c, proj


def _oracle_render(oracle, veach, arr, c, proj, W, H, spp):
    b = oracle.lbvh_build(veach["positions"], veach["face_position_indices"])
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    for s in range(1, spp + 1):
        oracle.pt_render_sample(b, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj),
                                c.pos, c.rotation(), acc, img, float(s))
    return acc, img


@pytest.mark.parametrize("W,H,spp", [(128, 96, 1), (200, 120, 4)])
def test_radiance_matches_checker(oracle, veach, W, H, spp):
    tracer, arr, c, proj = _setup(veach, W, H)
    fb = FrameBuffer(W, H, DEV)
    for _ in range(spp):
        tracer.Render(fb)
    torch.cuda.synchronize()
    assert tracer.GetCurrentAccumulatedSamplePerPixel() == spp + 1
    got = tracer.d_accum.cpu().numpy()
    ref, ref_img = _oracle_render(oracle, veach, arr, c, proj, W, H, spp)
    err = np.abs(got - ref).max(axis=2)
    frac_bad = float((err > TOL).mean())
    assert frac_bad <= 2e-3, f"{frac_bad:.5f} of the pixels differ by more than {TOL}"
    assert float(np.abs(got - ref).mean()) < 1e-4
    img = fb.ReadBack()
    assert (img[..., 3] == 255).all()
    assert float((np.abs(img.astype(int) - ref_img.astype(int)).max(axis=2) > 1).mean()) <= 2e-3


def test_both_trees_give_the_same_bits(veach):
    """The path tracer walks the 8-wide trace-side tree by default; on the binary tree as built it must
    accumulate exactly the same radiance (same hits, same shadow-ray answers, same random draws)."""
    W, H, spp = 240, 135, 3
    out = {}
    for wide in (True, False):
        tracer, *_ = _setup(veach, W, H)
        tracer.use_trace_tree = wide
        fb = FrameBuffer(W, H, DEV)
        for _ in range(spp):
            tracer.Render(fb)
        torch.cuda.synchronize()
        out[wide] = (tracer.d_accum.cpu().numpy().tobytes(), fb.ReadBack().tobytes())
    assert out[True] == out[False]


def test_row_split_gives_the_same_image(veach):
    W, H = 160, 100
    tracer, arr, c, proj = _setup(veach, W, H)
    fb = FrameBuffer(W, H, DEV)
    tracer.Render(fb)
    tracer.Render(fb)
    torch.cuda.synchronize()
    whole = tracer.d_accum.cpu().numpy().copy()
    tracer2, *_ = _setup(veach, W, H)
    fb2 = FrameBuffer(W, H, DEV)
    for _ in range(2):
        # both stripes of one sample, then the counter moves on once
        spp = tracer2.curr_accu_spp
        tracer2.Render(fb2, rows=(0, 37))
        tracer2.curr_accu_spp = max(spp, 1.0)
        tracer2._modified = False
        tracer2.Render(fb2, rows=(37, H))
    torch.cuda.synchronize()
    assert tracer2.d_accum.cpu().numpy().tobytes() == whole.tobytes()
    assert fb2.ReadBack().tobytes() == fb.ReadBack().tobytes()


def test_accumulation_restarts_on_change_and_stops_at_max(veach):
    W, H = 64, 48
    tracer, arr, c, proj = _setup(veach, W, H)
    fb = FrameBuffer(W, H, DEV)
    tracer.SetMaxSamplePerPixel(3.0)
    for _ in range(6):
        tracer.Render(fb)
    assert tracer.GetCurrentAccumulatedSamplePerPixel() == 4.0  # no-op once above the maximum
    tracer.SetMaxPathDepth(2)  # a parameter change restarts accumulation (ray_tracer_render.cu:561-574)
    tracer.Render(fb)
    assert tracer.GetCurrentAccumulatedSamplePerPixel() == 2.0


def test_render_without_lbvh_reports_and_returns(veach, capsys):
    tracer = RayTracer(DEV)
    tracer.Render(FrameBuffer(8, 8, DEV))
    assert "lbvh is not set" in capsys.readouterr().err


def _synthetic_scene():
    """A floor, a glossy tilted plate, a quad light, a sphere light stand-in and clusters of tiny
    triangles that share a Morton cell: leaves of a few dozen triangles (which get a node of their own
    in the trace-side tree) and of several hundred (tested 64 at a time)."""
    from raytracer_amd import synthetic as syn
    from raytracer_amd.data.mesh import ObjLight, QUAD, SPHERE
    f32 = np.float32
    tris, groups = [], []

    def quad(a, b, c, d):
        return [[a, b, c], [a, c, d]]

    groups.append(quad([-2, 0, -2], [-2, 0, 2], [2, 0, 2], [2, 0, -2]))                 # 0 floor
    groups.append(quad([-1.5, 0.1, -1.2], [-1.5, 1.4, -1.6], [0.2, 1.4, -1.6], [0.2, 0.1, -1.2]))  # 1 plate
    groups.append(quad([-0.4, 2.0, -0.4], [0.4, 2.0, -0.4], [0.4, 2.0, 0.4], [-0.4, 2.0, 0.4]))    # 2 quad light
    soup_pos, _ = syn.triangle_soup(660, half_extent=0.5, seed=77)
    soup = soup_pos.reshape(-1, 3, 3)
    cluster = []
    for k in range(12):                                     # twelve leaves of 30 tiny triangles
        centre = np.array([-1.2 + 0.2 * k, 0.3 + 0.05 * (k % 3), 0.6], f32)
        cluster += list(soup[30 * k:30 * k + 30] * f32(6e-4) + centre)
    cluster += list(soup[360:660] * f32(6e-4) + np.array([0.9, 0.5, -0.3], f32))  # one of 300
    groups.append([t.tolist() for t in cluster])            # 3 clusters
    big, _ = syn.triangle_soup(150, half_extent=0.25, seed=78)
    groups.append((big.reshape(-1, 3, 3) * f32(2.0) + np.array([-1.0, 0.0, -1.0], f32)).tolist())  # 4 shards
    positions, face_idx, normals, starts = [], [], [], []
    for g in groups:
        starts.append(len(face_idx))
        for t in g:
            t = np.asarray(t, f32)
            n = np.cross(t[1] - t[0], t[2] - t[0])
            n = n / max(np.linalg.norm(n), 1e-20)
            base = len(positions)
            positions += [t[0], t[1], t[2]]
            normals.append(n.astype(f32))
            face_idx.append([base, base + 1, base + 2])
    F = len(face_idx)
    G = len(groups)
    kd = np.array([[.6, .6, .6], [.1, .1, .1], [0, 0, 0], [.7, .3, .2], [.2, .5, .7]], f32)
    ks = np.array([[0, 0, 0], [.8, .8, .8], [0, 0, 0], [0, 0, 0], [.2, .2, .2]], f32)
    ns = np.array([1, 60, 1, 1, 12], f32)
    emission = np.zeros((G, 3), f32)
    emission[2] = [17, 12, 4]
    lights = [ObjLight(QUAD, np.array([17, 12, 4], f32), o=np.array([-0.4, 2.0, -0.4], f32),
                       u=np.array([0.8, 0, 0], f32), v=np.array([0, 0, 0.8], f32)),
              ObjLight(SPHERE, np.array([40, 40, 60], f32), o=np.array([1.5, 1.2, 1.0], f32), r=f32(0.1))]
    return dict(positions=np.asarray(positions, f32), normals=np.asarray(normals, f32),
                tex_coords=np.zeros((1, 2), f32),
                face_position_indices=np.asarray(face_idx, np.uint32),
                face_normal_indices=np.repeat(np.arange(F, dtype=np.uint32)[:, None], 3, axis=1),
                face_tex_coord_indices=np.zeros((F, 3), np.uint32),
                grp_start_face_indices=np.asarray(starts, np.uint32),
                mat_kd=kd, mat_ks=ks, mat_tr=np.zeros((G, 3), f32), mat_ni=np.ones(G, f32), mat_ns=ns,
                mat_emission=emission, obj_lights=lights,
                cam_eye=np.array([0.3, 1.3, 3.2], f32), cam_lookat=np.array([0, 0.4, 0], f32),
                cam_up=np.array([0, 1, 0], f32))


def test_quad_light_and_long_leaves(oracle):
    """A scene the reference's only data set does not cover: a quad light beside a sphere light, a glossy
    lobe, and leaves of 30 and of hundreds of triangles. The compacting kernel on the 8-wide tree, the
    one-pixel-per-lane kernel on the binary tree and the checker must agree."""
    scene = _synthetic_scene()
    W, H, spp = 160, 96, 3
    ref_tree = oracle.lbvh_build(scene["positions"], scene["face_position_indices"])
    runs = np.diff(np.append(ref_tree["compactedFaces"].astype(np.int64), scene["face_position_indices"].shape[0]))
    assert ((runs >= 4) & (runs <= 120)).sum() >= 8 and runs.max() > 120
    out = {}
    for wide in (True, False):
        tracer, arr, c, proj = _setup(scene, W, H)
        tracer.use_trace_tree = wide
        fb = FrameBuffer(W, H, DEV)
        for _ in range(spp):
            tracer.Render(fb)
        torch.cuda.synchronize()
        out[wide] = (tracer.d_accum.cpu().numpy(), fb.ReadBack())
    assert out[True][0].tobytes() == out[False][0].tobytes()
    assert out[True][1].tobytes() == out[False][1].tobytes()
    ref, ref_img = _oracle_render(oracle, scene, arr, c, proj, W, H, spp)
    got = out[True][0]
    assert float((np.abs(got - ref).max(axis=2) > TOL).mean()) <= 2e-3
    assert float(np.abs(got - ref).mean()) < 1e-4
    lit = got.max(axis=2) > 0.02
    assert 0.5 < float(lit.mean()) <= 1.0


def test_several_samples_in_one_launch_equal_separate_calls(veach):
    """Render(fb, samples=n) = n calls of Render(fb): every pixel folds its samples into the running mean
    in the same order, so the accumulation buffer and the surface are equal bit for bit; the counter
    stops at maxSamplePerPixel."""
    W, H = 192, 108
    tracer, *_ = _setup(veach, W, H)
    fb = FrameBuffer(W, H, DEV)
    for _ in range(5):
        tracer.Render(fb)
    torch.cuda.synchronize()
    one_by_one = (tracer.d_accum.cpu().numpy().tobytes(), fb.ReadBack().tobytes())
    tracer2, *_ = _setup(veach, W, H)
    fb2 = FrameBuffer(W, H, DEV)
    tracer2.Render(fb2, samples=2)
    tracer2.Render(fb2, samples=3)
    torch.cuda.synchronize()
    assert tracer2.GetCurrentAccumulatedSamplePerPixel() == 6
    assert (tracer2.d_accum.cpu().numpy().tobytes(), fb2.ReadBack().tobytes()) == one_by_one
    # the binary-tree path takes the same call
    tracer3, *_ = _setup(veach, W, H)
    tracer3.use_trace_tree = False
    fb3 = FrameBuffer(W, H, DEV)
    tracer3.Render(fb3, samples=5)
    torch.cuda.synchronize()
    assert (tracer3.d_accum.cpu().numpy().tobytes(), fb3.ReadBack().tobytes()) == one_by_one
    # never past the cap: two more samples fit under a cap of 7, a third call does nothing
    tracer2.max_sample_per_pixel = 7.0
    tracer2._prev_max_spp = 7.0
    tracer2.Render(fb2, samples=10)
    assert tracer2.GetCurrentAccumulatedSamplePerPixel() == 8
    tracer2.Render(fb2, samples=10)
    assert tracer2.GetCurrentAccumulatedSamplePerPixel() == 8


@pytest.mark.parametrize("world,use_tree", [(2, True), (3, True), (2, False)])
def test_compact_stripes_hold_the_rows_of_the_single_gpu_frame(veach, world, use_tree):
    """What rank r of `world` GPUs renders - its round-robin 16-row stripes, compact, accumulated in a
    compact buffer - is byte for byte the same rows of the frame one GPU renders: the random stream is
    keyed by the image pixel. H is no multiple of 16 (last stripe partial); three samples, once as three
    calls and once as one launch."""
    from raytracer_amd import multi_gpu
    W, H, spp = 200, 120 - 6, 3
    tracer, arr, c, proj = _setup(veach, W, H)
    tracer.use_trace_tree = use_tree
    fb = FrameBuffer(W, H, DEV)
    for _ in range(spp):
        tracer.Render(fb)
    torch.cuda.synchronize()
    want_acc, want_img = tracer.d_accum.cpu().numpy(), fb.ReadBack()
    per = multi_gpu.stripes_per_rank(H, world)
    for batched in (False, True):
        blocks = torch.zeros((world, per, 16, W, 4), dtype=torch.uint8, device=DEV)
        for r in range(world):
            t2, _, _, _ = _setup(veach, W, H)
            t2.use_trace_tree = use_tree
            if batched:
                t2.Render(fb, samples=spp, stripes=(r, world), compact_out=blocks[r])
            else:
                for _ in range(spp):
                    t2.Render(fb, stripes=(r, world), compact_out=blocks[r])
            torch.cuda.synchronize()
            assert t2.GetCurrentAccumulatedSamplePerPixel() == spp + 1
            acc = t2.d_accum.cpu().numpy()
            assert acc.shape == (16 * per, W, 3)
            for j, s in enumerate(multi_gpu.stripes_of(H, world, r)):
                rows = min(16, H - 16 * s)
                assert acc[16 * j:16 * j + rows].tobytes() == want_acc[16 * s:16 * s + rows].tobytes()
        out = torch.zeros((H, W, 4), dtype=torch.uint8, device=DEV)
        multi_gpu.unstripe(blocks, world, out)
        assert out.cpu().numpy().tobytes() == want_img.tobytes()


def test_stated_size_of_the_second_configuration_matches_checker(oracle, veach):
    """The example scene at 1920 x 1080 with 16 calls of Render(Scene) (the sequence of
    app/rt_lbvh.cpp:239-296): a 16-row band through the lights and the floor is compared with the CPU
    checker's accumulation of the same rows, within 1e-3 absolute."""
    W, H, spp = 1920, 1080, 16
    tracer, arr, c, proj = _setup(veach, W, H)
    fb = FrameBuffer(W, H, DEV)
    for _ in range(spp):
        tracer.Render(fb)
    torch.cuda.synchronize()
    assert tracer.GetCurrentAccumulatedSamplePerPixel() == spp + 1
    rows = (H // 2 - 8, H // 2 + 8)
    b = oracle.lbvh_build(veach["positions"], veach["face_position_indices"])
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    for s in range(1, spp + 1):
        oracle.pt_render_sample(b, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj),
                                c.pos, c.rotation(), acc, img, float(s), rows=rows)
    got = tracer.d_accum[rows[0]:rows[1]].cpu().numpy()
    ref = acc[rows[0]:rows[1]]
    assert float(ref.max()) > 0.2, "the band shows nothing"
    err = np.abs(got - ref).max(axis=2)
    assert float((err > TOL).mean()) <= 2e-3
    assert float(np.abs(got - ref).mean()) < 1e-4
    gimg = fb.ReadBack()[rows[0]:rows[1]]
    assert float((np.abs(gimg.astype(int) - img[rows[0]:rows[1]].astype(int)).max(axis=2) > 1).mean()) <= 2e-3
