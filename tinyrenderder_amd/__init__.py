"""tinyrenderder_amd — MI355X-native tile rasterizer behind the reference's rasterize()/IShader surface.

Only the hot path of SURVEY.md §8 lives here: csrc/ (HIP kernels + the C ABI of include/trgl.h),
api.py (the Python host mirror over ctypes) and scenes.py (seeded synthetic workloads).
"""
from . import scenes  # noqa: F401

__all__ = ["scenes"]
