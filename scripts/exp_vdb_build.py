"""Where the wall time of DepthBoxVDB.BuildFrom goes (Kingsnake shape), cold and warm."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raytracer_amd import synthetic as syn
from raytracer_amd.data.transfer_function import README_KINGSNAKE_TF, TransferFunction
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume

dev = "cuda:0"
tf = TransferFunction(README_KINGSNAKE_TF)
vol = Volume(syn.kingsnake_volume_torch((1024, 1024, 795), dev), dev)
torch.cuda.synchronize()
for i in range(4):
    vdb = DepthBoxVDB(dev)
    t0 = time.perf_counter()
    vdb.BuildFrom(vol, tf.empty_scalar_ranges, apron_wid=1, log2_dims=(5, 4, 3), use_depth_box=True)
    torch.cuda.synchronize()
    print(f"build {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", {k: round(v, 3) for k, v in vdb.timings.items()})
