"""The calls of the N > 1 path made on the GPU with a one-rank RCCL group: the broadcast of a built tree
and of a VDB, the stripe gather, and closest hits / a DVR frame on the structures afterwards. (Two
ranks are rehearsed with gloo on the CPU in test_multi_gpu_cpu.py; a one-GPU box cannot hold two.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DEV = "cuda:0"


@pytest.fixture(scope="module")
def group():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    except Exception as e:  # no RCCL on this box: nothing to rehearse
        pytest.skip(f"RCCL group could not be created: {e}")
    yield dist
    dist.destroy_process_group()


def test_broadcast_tree_and_trace(group):
    from raytracer_amd import multi_gpu, synthetic as syn
    from raytracer_amd.lbvh import LBVH
    from raytracer_amd.ray_tracer import closest_hit
    pos, tri = syn.triangle_soup(50_000)
    rays = syn.incoherent_rays(20_000)
    bvh = LBVH(DEV)
    bvh.BuildFrom(pos, tri)
    before = [x.cpu().numpy() for x in closest_hit(bvh, rays)]
    multi_gpu.broadcast_lbvh(bvh, src=0)
    group.barrier()
    after = [x.cpu().numpy() for x in closest_hit(bvh, rays)]
    for a, b in zip(before, after):
        assert a.tobytes() == b.tobytes()
    assert multi_gpu.split_range(20_000, 1, 0) == (0, 20_000)


def test_broadcast_vdb_and_gather_stripes(group):
    from raytracer_amd import multi_gpu, synthetic as syn
    from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
    from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable
    from vol_util import HEAD_TF, dvr_setup
    dim = (64, 128, 128)
    vdb = DepthBoxVDB(DEV)
    vdb.BuildFrom(Volume(syn.head_volume(dim), DEV), HEAD_TF.empty_scalar_ranges, apron_wid=1,
                  log2_dims=(4, 3, 3), use_depth_box=True)
    multi_gpu.broadcast_vdb(vdb, src=0)
    W, H = 160, 120
    w2s, c, proj = dvr_setup(dim, (2, 1, 1), W, H, 1.0, 9)
    caster = RayCaster(DEV)
    caster.SetWorldToScene(w2s)
    caster.SetDepthBoxVDB(vdb)
    caster.SetTransferFunctionTexture(TransferFunctionTable(HEAD_TF.dat, DEV))
    caster.SetProjection(proj)
    caster.SetCameraToWorld(c.pos, c.rotation())
    whole, striped = FrameBuffer(W, H, DEV), FrameBuffer(W, H, DEV)
    caster.RenderDepthBoxVDB(whole)
    caster.RenderDepthBoxVDB(striped, stripes=(0, 1))
    multi_gpu.gather_stripes(striped.rgba, 1, 0, 0, {})
    torch.cuda.synchronize()
    assert whole.ReadBack().tobytes() == striped.ReadBack().tobytes()
    assert (whole.ReadBack()[..., 3] > 0).mean() > 0.05
