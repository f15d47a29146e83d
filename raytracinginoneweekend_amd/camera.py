"""Camera with the reference's interface (camera.nim).

initCamera runs once on the host in binary64 exactly as camera.nim:12-31 does; the per-sample
getRay of camera.nim:34-39 runs on the GPU.
"""
import math

from . import _lib
from .basetypes import Point, Vec3, cross, unitVector


class Camera:
    """camera.nim:4-10"""

    __slots__ = ("origin", "lowerLeftCorner", "horizontal", "vertical", "lensRadius", "u", "v", "w")

    def to_struct(self, binary32=False):
        """The eight fields for rtow_set_camera_f64 (as computed, binary64), or rounded to binary32
        for rtow_set_camera."""
        s = _lib.CameraStruct() if binary32 else _lib.CameraStruct64()
        for name, field in (("origin", self.origin), ("lower_left", self.lowerLeftCorner),
                            ("horizontal", self.horizontal), ("vertical", self.vertical),
                            ("u", self.u), ("v", self.v), ("w", self.w)):
            arr = getattr(s, name)
            for i in range(3):
                arr[i] = field[i]
        s.lens_radius = self.lensRadius
        return s


def initCamera(lookFrom, lookAt, vup, vfov, aspectRatio, aperture=0.0, focusDist=None):
    if focusDist is None:
        raise TypeError("initCamera: focusDist is required (camera.nim:12-14 gives it no default)")
    theta = float(vfov) * (math.pi / 180.0)  # degToRad
    h = math.tan(theta / 2.0)
    viewHeight = 2.0 * h
    viewWidth = aspectRatio * viewHeight

    c = Camera()
    c.w = unitVector(Vec3(*(lookFrom - lookAt)))
    c.u = unitVector(cross(vup, c.w))
    c.v = cross(c.w, c.u)

    c.origin = lookFrom
    c.horizontal = focusDist * viewWidth * c.u
    c.vertical = focusDist * viewHeight * c.v
    c.lowerLeftCorner = (c.origin - Point(*c.horizontal) / 2.0 - Point(*c.vertical) / 2.0
                         - focusDist * Point(*c.w))
    c.lensRadius = aperture / 2.0
    return c


def getRay(c, s, t, **kw):
    """getRay*(c: Camera, s, t: float): Ray (camera.nim:34-39), evaluated on the GPU (raytracer.getRay)."""
    from . import raytracer
    return raytracer.getRay(c, s, t, **kw)
