"""bench.py with two ranks on one GPU: the launch line of the multi-GPU runs with KK_BENCH_REHEARSAL=1
(gloo, both ranks on GPU 0). Checks that the N > 1 path runs through and prints its one JSON line; the
numbers of such a run mean nothing."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_rehearsal():
    env = dict(os.environ, KK_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--tris", "200000", "--rays", "65536", "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["dvr"]["n_gpus"] == 2 and out["dvr"]["fps"] > 0
    assert "lbvh_broadcast_ms" in out and "vdb_broadcast_ms" in out["dvr"]
    # the frame put together from two ranks' stripes is the frame one GPU renders, byte for byte
    assert out["dvr"]["gathered_frame_equals_single_gpu"] is True
    assert out["dvr_head_1080p"]["gathered_frame_equals_single_gpu"] is True
    # the path-traced frame shared between the two ranks: scene + tree broadcast, stripes gathered per call
    assert out["path_trace"]["n_gpus"] == 2 and out["path_trace"]["ms_per_frame"] > 0
    assert out["path_trace"]["gathered_frame_equals_single_gpu"] is True
    assert "gather_us_per_frame" in out["dvr"] and out["dvr"]["render_only_ms_per_frame"] > 0
    # ... and so is the last frame of the loop that keeps two frames in flight on every rank
    assert out["dvr"]["fps_two_frames_in_flight"] > 0
    assert out["dvr"]["two_in_flight_gathered_frame_equals_single_gpu"] is True
    assert out["dvr_head_1080p"]["two_in_flight_gathered_frame_equals_single_gpu"] is True
