"""Headless counterparts of the reference's two applications (app/rt_lbvh.cpp, app/dvr_dpbx_vdb.cpp):
same command-line options, the frame goes to a PPM / NPY file instead of a window."""
import numpy as np


def write_image(path: str, rgba: np.ndarray) -> None:
    """rgba [H, W, 4] uint8, row 0 = bottom row (GL convention) -> binary PPM (top row first) or .npy."""
    if path.endswith(".npy"):
        np.save(path, rgba)
        return
    rgb = np.ascontiguousarray(rgba[::-1, :, :3])
    with open(path, "wb") as f:
        f.write(f"P6\n{rgb.shape[1]} {rgb.shape[0]}\n255\n".encode())
        f.write(rgb.tobytes())
