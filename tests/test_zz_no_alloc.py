"""kk_sort_pairs_u32_static, the sort for callers that cannot allocate device memory: its histograms,
tile counters and look-back rows live in a device array inside the library. Through the C ABI it must
give exactly what kk_sort_pairs_u32 gives, on two calls in a row (the second with a smaller input, so
that it meets the first call's leftovers in the shared array), and reject sizes the array cannot hold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from raytracer_amd import _lib  # noqa: E402

DEV = "cuda:0"
MAX_N = 1 << 22  # KK_SORT_STATIC_MAX_N


def _buffers(keys, vals):
    ka = torch.from_numpy(keys.view(np.int32).copy()).to(DEV)
    va = torch.empty_like(ka) if vals is None else torch.from_numpy(vals.view(np.int32).copy()).to(DEV)
    return ka, torch.empty_like(ka), va, torch.empty_like(ka)


def _sort_workspace(keys, vals=None):
    l = _lib.lib()
    n = keys.shape[0]
    ka, kb, va, vb = _buffers(keys, vals)
    ws = torch.empty(l.kk_sort_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    _lib.check(l.kk_sort_pairs_u32(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(),
                                   1 if vals is None else 0, n, ws.data_ptr(), ws.numel(),
                                   _lib.current_stream_ptr()), "kk_sort_pairs_u32")
    torch.cuda.synchronize()
    return ka.cpu().numpy().view(np.uint32), va.cpu().numpy().view(np.uint32)


def _sort_static(keys, vals=None):
    l = _lib.lib()
    n = keys.shape[0]
    ka, kb, va, vb = _buffers(keys, vals)  # inputs and outputs are still the caller's
    # the array's address is looked up (and the code object loaded) outside the measured call, as a caller
    # under graph capture would do it
    _lib.check(l.kk_sort_static_init(), "kk_sort_static_init")
    torch.cuda.synchronize()
    # free device memory as the driver reports it (hipMemGetInfo): a hipMalloc inside the library would
    # show here, which torch's own allocator statistics would not see
    free_before, _ = torch.cuda.mem_get_info()
    before = torch.cuda.memory_allocated()
    _lib.check(l.kk_sort_pairs_u32_static(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(),
                                          1 if vals is None else 0, n, _lib.current_stream_ptr()),
               "kk_sort_pairs_u32_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    _lib.check(l.kk_sort_static_status(_lib.current_stream_ptr()), "kk_sort_static_status")
    return ka.cpu().numpy().view(np.uint32), va.cpu().numpy().view(np.uint32)


def test_static_sort_equals_workspace_sort_twice_in_a_row():
    rng = np.random.default_rng(2024)
    # the first call fills many status rows; the second, smaller one runs over their leftovers
    _sort_static(np.arange(4096, dtype=np.uint32)[::-1].copy())  # (first launch: the runtime's own set-up)
    for n, with_vals in ((3_000_001, False), (70_003, True), (MAX_N, False), (1, False)):
        keys = rng.integers(0, 1 << 30, n, dtype=np.uint64).astype(np.uint32)
        keys[::7] = keys[0]  # duplicates: stability shows
        vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) if with_vals else None
        k_ref, v_ref = _sort_workspace(keys, vals)
        k, v = _sort_static(keys, vals)
        np.testing.assert_array_equal(k, k_ref)
        np.testing.assert_array_equal(v, v_ref)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(k, keys[order])
        np.testing.assert_array_equal(v, order.astype(np.uint32) if vals is None else vals[order])


def test_static_sort_rejects_sizes_beyond_its_scratch():
    l = _lib.lib()
    n = MAX_N + 1
    k = torch.zeros(n, dtype=torch.int32, device=DEV)
    kb, va, vb = torch.empty_like(k), torch.empty_like(k), torch.empty_like(k)
    rc = l.kk_sort_pairs_u32_static(k.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, n,
                                    _lib.current_stream_ptr())
    assert rc == -1
    assert b"KK_SORT_STATIC_MAX_N" in l.kk_last_error()
    rc = l.kk_sort_pairs_u32_static(k.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, 0,
                                    _lib.current_stream_ptr())
    assert rc == -1
