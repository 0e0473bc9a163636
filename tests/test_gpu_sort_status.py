"""The radix sort's look-back can give up (a spin limit stands in for a forward-progress guarantee HIP does
not give). Every caller of the sort must then say so and must not use the order it got: the LBVH build is
not complete, the closest-hit kernel traces in input order (same hits), the tile order falls back to the
launch order, and the two sort entry points report KK_ERR_DEVICE. The error word is planted by the
library's test hook (kk_debug_sort_force_error), which raises it after a sort exactly as a look-back
that gave up would. Also: ticketed mode (kk_sort_set_ticketed) sorts and builds the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from raytracer_amd import _lib, synthetic as syn  # noqa: E402
from raytracer_amd.lbvh import LBVH  # noqa: E402
from raytracer_amd.ray_tracer import closest_hit, closest_hit_status  # noqa: E402

DEV = "cuda:0"
KK_ERR_DEVICE = -4


@pytest.fixture
def forced_error():
    l = _lib.lib()
    l.kk_debug_sort_force_error(1)
    yield
    l.kk_debug_sort_force_error(0)


def _sort(n, seed=5):
    l = _lib.lib()
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint64).astype(np.uint32)
    ka = torch.from_numpy(keys.view(np.int32)).to(DEV)
    kb, va, vb = torch.empty_like(ka), torch.empty_like(ka), torch.empty_like(ka)
    ws = torch.empty(l.kk_sort_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    _lib.check(l.kk_sort_pairs_u32(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, n,
                                   ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()), "kk_sort_pairs_u32")
    return keys, ka, va, ws


def test_error_code_value():
    # the header's KK_ERR_DEVICE, as the status calls below return it
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "kouek_hip.h")).read()
    assert int(re.search(r"KK_ERR_DEVICE\s*=\s*(-?\d+)", hdr).group(1)) == KK_ERR_DEVICE


def test_sort_entry_points_report_a_given_up_look_back(forced_error):
    l = _lib.lib()
    n = 100_003
    keys, ka, va, ws = _sort(n)
    assert l.kk_sort_workspace_status(ws.data_ptr(), n, _lib.current_stream_ptr()) == KK_ERR_DEVICE
    assert b"look-back gave up" in l.kk_last_error()
    kb, vb = torch.empty_like(ka), torch.empty_like(ka)
    _lib.check(l.kk_sort_pairs_u32_static(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, n,
                                          _lib.current_stream_ptr()), "kk_sort_pairs_u32_static")
    assert l.kk_sort_static_status(_lib.current_stream_ptr()) == KK_ERR_DEVICE
    l.kk_debug_sort_force_error(0)
    keys, ka, va, ws = _sort(n)
    assert l.kk_sort_workspace_status(ws.data_ptr(), n, _lib.current_stream_ptr()) == 0
    _lib.check(l.kk_sort_pairs_u32_static(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, n,
                                          _lib.current_stream_ptr()), "kk_sort_pairs_u32_static")
    assert l.kk_sort_static_status(_lib.current_stream_ptr()) == 0


def test_build_is_not_complete_after_a_given_up_sort(forced_error, capsys):
    pos, tri = syn.triangle_soup(5_000)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    assert not bvh.IsComplete()
    assert "look-back gave up" in capsys.readouterr().err


def test_closest_hit_traces_in_input_order_and_says_so():
    l = _lib.lib()
    pos, tri = syn.triangle_soup(20_000)
    rays = torch.from_numpy(syn.incoherent_rays(8192)).to(DEV)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    assert bvh.IsComplete()
    t0, b0, p0 = closest_hit(bvh, rays, check=True)
    R = rays.shape[0]
    out = (torch.empty_like(t0), torch.empty_like(b0), torch.empty_like(p0),
           torch.empty(l.kk_closest_hit_accel_workspace_bytes(R), dtype=torch.uint8, device=DEV))
    l.kk_debug_sort_force_error(1)
    try:
        t1, b1, p1 = closest_hit(bvh, rays, out=out)
        with pytest.raises(_lib.KoukeError, match="input order"):
            closest_hit_status(out[3], R)
        assert torch.equal(p1, p0) and torch.equal(t1, t0) and torch.equal(b1, b0)
        # the order array is garbage as far as the kernel knows: scramble it (everything in front of the
        # sort's own workspace and the status word, which close the workspace), then ask for its reuse
        tail = int(l.kk_sort_workspace_bytes(R)) + 256
        out[3][:out[3].numel() - tail].random_(0, 255)
        out[0].fill_(-1.0)
        out[2].fill_(7)
        t1, b1, p1 = closest_hit(bvh, rays, out=out, reuse_order=True)
        with pytest.raises(_lib.KoukeError, match="input order"):
            closest_hit_status(out[3], R)
        assert torch.equal(p1, p0) and torch.equal(t1, t0) and torch.equal(b1, b0)
        out[0].fill_(-1.0)
        out[2].fill_(7)
    finally:
        l.kk_debug_sort_force_error(0)
    t2, b2, p2 = closest_hit(bvh, rays, out=out)  # a good sort again: status clear
    closest_hit_status(out[3], R)
    assert torch.equal(p2, p0) and torch.equal(t2, t0) and torch.equal(b2, b0)


def test_tile_order_falls_back_to_launch_order(forced_error):
    l = _lib.lib()
    n = 32_400
    cost = torch.randint(0, 60000, (n,), dtype=torch.int32, device=DEV)
    order = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    ws = torch.zeros(l.kk_dvr_tile_order_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    _lib.check(l.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                   _lib.current_stream_ptr()), "kk_dvr_tile_order")
    assert l.kk_dvr_tile_order_status(ws.data_ptr(), n, _lib.current_stream_ptr()) == KK_ERR_DEVICE
    assert torch.equal(order, torch.arange(n, dtype=torch.int32, device=DEV))
    l.kk_debug_sort_force_error(0)
    _lib.check(l.kk_dvr_tile_order(cost.data_ptr(), n, order.data_ptr(), ws.data_ptr(), ws.numel(),
                                   _lib.current_stream_ptr()), "kk_dvr_tile_order")
    assert l.kk_dvr_tile_order_status(ws.data_ptr(), n, _lib.current_stream_ptr()) == 0
    got = order.cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(n))
    assert not np.array_equal(got, np.arange(n))


def test_ticketed_mode_gives_the_same_sort_and_the_same_tree():
    l = _lib.lib()
    pos, tri = syn.triangle_soup(300_000)
    plain = LBVH(DEV)
    plain.BuildFrom(pos, tri)
    ref = plain.to_host()
    keys, ka0, va0, _ = _sort(1_000_003, seed=9)
    torch.cuda.synchronize()
    old = l.kk_sort_set_ticketed(1)
    try:
        assert old == 0
        keys, ka1, va1, ws = _sort(1_000_003, seed=9)
        assert l.kk_sort_workspace_status(ws.data_ptr(), 1_000_003, _lib.current_stream_ptr()) == 0
        assert torch.equal(ka0, ka1) and torch.equal(va0, va1)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(va1.cpu().numpy().view(np.uint32), order.astype(np.uint32))
        ticketed = LBVH(DEV)
        ticketed.BuildFrom(pos, tri)
        assert ticketed.IsComplete()
        got = ticketed.to_host()
    finally:
        l.kk_sort_set_ticketed(0)
    assert got["M"] == ref["M"]
    for k in ("faces", "compactedFaces", "sortedMortons", "uniqueMortons"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["internalNodes"].tobytes() == ref["internalNodes"].tobytes()
