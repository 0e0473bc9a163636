"""MI355X (gfx950) implementation of the two data-parallel paths of Kouek/RayTracer.

The GPU work lives in the C-ABI library built from raytracer_amd/csrc (lib/libkouek_hip.so, declared in
include/kouek_hip.h). The modules here are the host-side mirror of the reference's classes for Python
callers: they own device memory through torch tensors and pass raw pointers across the C ABI.
"""
from ._lib import lib, build_library, KoukeError  # noqa: F401
from .lbvh import LBVH  # noqa: F401
