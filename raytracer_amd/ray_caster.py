"""Host-side mirror of kouek::RayCaster::RayCaster (include/ray_caster/ray_caster.h:16-89) for the
Depth-Box VDB path. The CUDA surface becomes a FrameBuffer, the transfer-function texture a float4[256]
table in device memory."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch

from . import _lib
from .dpbx_vdb import DepthBoxVDB
from .ray_tracer import FrameBuffer  # noqa: F401  (same headless framebuffer)


class TransferFunctionTable:
    """Replaces the 1-D float4 CUDA texture of the transfer function (app/dvr_dpbx_vdb.cpp:246-269)."""

    def __init__(self, table, device="cuda:0"):
        t = np.ascontiguousarray(table, np.float32).reshape(256, 4)
        self.tensor = torch.from_numpy(t).to(device)


class RayCaster:
    ErrTag = "[RayCaster Error]"

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self._params = _lib.DvrParams()
        self._params.max_alpha = 0.9
        self._params.sampling_rate = 3.0
        self._frame = _lib.DvrFrame()
        self.vdb = None
        self.tf = None
        self.volume = None
        self._display_lev = 0
        self._display_leaf_layer = 0
        self._adaptive = False
        self._tile_states = {}
        self._tile_state = None

    def SetAdaptiveTileOrder(self, on: bool = True):
        """Start each Scene frame's 16 x 16 tiles (whole frames, row ranges, and one GPU's compact
        stripes) in the order of what they took in the frame before, the costly ones first. The image
        is the same; the frame is shorter because the longest rays no longer start last. The state belongs to this object's sequence of frames on one stream: frames
        in flight on several streams take a RayCaster each. No counterpart in the reference, whose
        launch order is the tiles' own (src/ray_caster/ray_caster_dpbx_vdb.cu:1027-1034)."""
        self._adaptive = bool(on)
        self._tile_states = {}
        self._tile_state = None

    def _tiles(self, key, n):
        """Cost, order and sort workspace of the adaptive tile order for a launch of n tiles; `key` names
        the launch (size, rows or stripes); every launch keeps its own tile times."""
        st = self._tile_states.get(key)
        if st is None:
            if len(self._tile_states) >= 8:  # an application alternates between a few launches at most
                self._tile_states.clear()
            ws = int(_lib.lib().kk_dvr_tile_order_workspace_bytes(n))
            st = {"key": key, "n": n, "have": False,
                  "cost": torch.zeros(n, dtype=torch.int32, device=self.device),
                  "order": torch.zeros(n, dtype=torch.int32, device=self.device),
                  "ws": torch.zeros(ws, dtype=torch.uint8, device=self.device)}
            self._tile_states[key] = st
        self._tile_state = st  # the launch rendered last
        return st

    def _render_ordered(self, fn, st, *args):
        """One ordered launch (fn's arguments up to the tile arrays in args) and the sort of its tile times
        for the next one, on the current stream."""
        with torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr()
            _lib.check(fn(*args, st["cost"].data_ptr(), st["order"].data_ptr() if st["have"] else None, stream),
                       "kk_dvr_render_ordered")
            _lib.check(_lib.lib().kk_dvr_tile_order(
                st["cost"].data_ptr(), st["n"], st["order"].data_ptr(), st["ws"].data_ptr(),
                st["ws"].numel(), stream), "kk_dvr_tile_order")
        st["have"] = True

    def CheckTileOrder(self) -> None:
        """Waits for the current stream and raises KoukeError when the sort behind the adaptive tile
        order of the launch rendered last gave up (the library then keeps the launch order for the next
        frame, so every tile is still rendered; only the frame-time gain is lost)."""
        st = self._tile_state
        if st is None or not st["have"]:
            return
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().kk_dvr_tile_order_status(st["ws"].data_ptr(), st["n"],
                                                           _lib.current_stream_ptr()),
                       "kk_dvr_tile_order_status")

    @staticmethod
    def _fill(arr, value):
        for i, x in enumerate(np.asarray(value, np.float32).reshape(-1)):
            arr[i] = float(x)

    def SetRAWVolume(self, volume):
        """volume: dpbx_vdb.Volume (the dense uint8 / uint16 volume, [dz, dy, dx], on this device)."""
        self.volume = volume

    def RenderRAWVolume(self, fb: FrameBuffer, target: str = "Scene", rows=None, counters=None) -> None:
        """RayCaster::RenderRAWVolume (src/ray_caster/ray_caster_raw_volume.cu:147-246): the marcher
        without empty-space skipping; target "Scene" or "AABB" (RAWVolumeRenderTarget)."""
        if self.volume is None or self.tf is None:
            print(f"{self.ErrTag} volTex or tfTex are not set.", file=sys.stderr)
            return
        t = self.volume.tensor
        sfx = "u8" if t.element_size() == 1 else "u16"
        dz, dy, dx = t.shape
        y0, y1 = rows if rows is not None else (0, fb.height)
        with torch.cuda.device(self.device):
            _lib.check(getattr(_lib.lib(), f"kk_raw_render_{sfx}")(
                t.data_ptr(), dx, dy, dz, self.tf.tensor.data_ptr(), C.byref(self._params),
                C.byref(self._frame), ("Scene", "AABB").index(target), fb.rgba.data_ptr(), fb.width,
                fb.height, int(y0), int(y1), counters.data_ptr() if counters is not None else None,
                _lib.current_stream_ptr()), "kk_raw_render")

    def SetDepthBoxVDB(self, vdb: DepthBoxVDB):
        self.vdb = vdb

    def SetDepthBoxVDBDisplayLevel(self, lev: int):
        """Level whose nodes the AABBs target colours: 0 = leaf bricks ... rootLev."""
        self._display_lev = int(lev) & 0xFF

    def SetDepthBoxVDBDisplayLeafLayer(self, layer: int):
        """Which leaf along each ray (0 = first) the Depths / PositionsAfterDepthSkip targets show."""
        self._display_leaf_layer = int(layer) & 0xFF

    def SetTransferFunctionTexture(self, tf: TransferFunctionTable):
        self.tf = tf

    def SetMaximumAlpha(self, a: float):
        self._params.max_alpha = float(a)

    def SetSamplingRate(self, r: float):
        self._params.sampling_rate = float(r)

    def SetWorldToScene(self, w2s):
        self._fill(self._params.w2s, w2s)

    def SetCameraToWorld(self, eye_pos, eye_rot):
        self._fill(self._frame.eye_pos, eye_pos)
        self._fill(self._frame.eye_rot, eye_rot)

    def SetProjection(self, proj):
        from .data.camera import inverse_projective
        self._fill(self._params.inv_proj, inverse_projective(proj))

    def SetShading(self, ka, kd, ks, shiness, light_pos_in_volume_space, light_col):
        p = self._params
        p.use_shading, p.ka, p.kd, p.ks, p.shiness = 1, float(ka), float(kd), float(ks), float(shiness)
        self._fill(p.light_pos, light_pos_in_volume_space)
        self._fill(p.light_col, light_col)

    def UnsetShading(self):
        self._params.use_shading = 0

    DPBXVDB_RENDER_TARGETS = ("Scene", "AABBs", "Depths", "PositionsAfterDepthSkip")

    def RenderDepthBoxVDB(self, fb: FrameBuffer, rows=None, counters=None, stripes=None,
                          target: str = "Scene", compact_out=None) -> None:
        """One frame into fb. `rows` = (y0, y1) renders those rows only; `stripes` = (first, stride)
        renders the 16-row stripes first, first + stride, ... (one GPU's share; Scene only); with
        `compact_out` (uint8 tensor [per, 16, W, 4]) those stripes are written one after the other into
        it instead of into fb (the block a rank sends to the gathering rank). `target` is a
        DPBXVDBRenderTarget (include/ray_caster/ray_caster.h:85)."""
        if self.vdb is None or self.tf is None or not self.vdb.IsComplete():
            print(f"{self.ErrTag} vdb or tfTex are not set.", file=sys.stderr)
            return
        sfx = "u8" if self.vdb.voxel_bytes == 1 else "u16"
        if target != "Scene":
            y0, y1 = rows if rows is not None else (0, fb.height)
            with torch.cuda.device(self.device):
                _lib.check(getattr(_lib.lib(), f"kk_dvr_render_target_{sfx}")(
                    C.byref(self.vdb.GetDeviceData()), self.tf.tensor.data_ptr(), C.byref(self._params),
                    C.byref(self._frame), self.DPBXVDB_RENDER_TARGETS.index(target), self._display_lev,
                    self._display_leaf_layer, fb.rgba.data_ptr(), fb.width, fb.height, int(y0), int(y1),
                    _lib.current_stream_ptr()), "kk_dvr_render_target")
            return
        out_ptr = fb.rgba.data_ptr()
        if stripes is not None and compact_out is not None:
            fn, a, b = getattr(_lib.lib(), f"kk_dvr_render_stripes_compact_{sfx}"), stripes[0], stripes[1]
            need = _lib.lib().kk_stripes_per_rank(fb.height, stripes[1]) * 16 * fb.width * 4
            if compact_out.numel() * compact_out.element_size() < need:
                raise _lib.KoukeError("compact_out is smaller than this rank's stripes")
            out_ptr = compact_out.data_ptr()
        elif stripes is not None:
            fn, a, b = getattr(_lib.lib(), f"kk_dvr_render_stripes_{sfx}"), stripes[0], stripes[1]
        else:
            fn = getattr(_lib.lib(), f"kk_dvr_render_{sfx}")
            a, b = rows if rows is not None else (0, fb.height)
            if self._adaptive and counters is None and b > a:
                n = int(_lib.lib().kk_dvr_tile_count(fb.width, int(b) - int(a)))
                self._render_ordered(
                    getattr(_lib.lib(), f"kk_dvr_render_ordered_{sfx}"),
                    self._tiles(("rows", fb.width, fb.height, int(a), int(b)), n),
                    C.byref(self.vdb.GetDeviceData()), self.tf.tensor.data_ptr(), C.byref(self._params),
                    C.byref(self._frame), out_ptr, fb.width, fb.height, int(a), int(b))
                return
        if self._adaptive and counters is None and stripes is not None and compact_out is not None:
            n = int(_lib.lib().kk_dvr_stripe_tile_count(fb.width, fb.height, int(a), int(b)))
            if n > 0:
                self._render_ordered(
                    getattr(_lib.lib(), f"kk_dvr_render_stripes_compact_ordered_{sfx}"),
                    self._tiles(("stripes", fb.width, fb.height, int(a), int(b)), n),
                    C.byref(self.vdb.GetDeviceData()), self.tf.tensor.data_ptr(), C.byref(self._params),
                    C.byref(self._frame), out_ptr, fb.width, fb.height, int(a), int(b))
                return
        with torch.cuda.device(self.device):
            _lib.check(fn(C.byref(self.vdb.GetDeviceData()), self.tf.tensor.data_ptr(),
                          C.byref(self._params), C.byref(self._frame), out_ptr, fb.width,
                          fb.height, int(a), int(b),
                          counters.data_ptr() if counters is not None else None,
                          _lib.current_stream_ptr()), "kk_dvr_render")
