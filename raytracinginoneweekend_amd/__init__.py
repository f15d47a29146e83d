"""MI355X-native renderer behind the API of Vindaar/rayTracingInOneWeekend.

The host modules keep the reference's module and proc names (basetypes, hitrecord, hittables,
camera, raytracer); `raytracer.render` flattens the scene and calls the HIP library through the
C ABI declared in include/rtow.h.  There is no CPU rendering path in this package.
"""
from . import basetypes, camera, hitrecord, hittables, nimrandom, raytracer  # noqa: F401
from .raytracer import Renderer, render  # noqa: F401

__all__ = ["basetypes", "camera", "hitrecord", "hittables", "nimrandom", "raytracer", "Renderer", "render"]
