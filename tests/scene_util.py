"""The veach-mis path-tracing set-up used by several tests (kept in the package so the benchmark can
use it too)."""
from raytracer_amd.data.scene_setup import veach_arrays, veach_camera  # noqa: F401
