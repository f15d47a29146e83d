"""The scene container with the reference's interface (hittables.nim:3-73).

The reference grows a raw buffer by hand; a Python list gives the same observable behaviour
(append order = list index, which decides ties in the closest-hit scan, hittables.nim:84-88).
`flatten` produces the structure-of-arrays form the C ABI uploads.
"""
import numpy as np

from .hitrecord import Sphere

htSphere = 0


class Hittable:
    """hittables.nim:4-8: a variant with a single member today."""

    def __init__(self, kind=htSphere, hSphere=None):
        self.kind = kind
        self.hSphere = hSphere


class HittablesList:
    def __init__(self, size=8):
        self.size = 8 if size < 8 else size  # reserved slots, informational only
        self._data = []

    def __len__(self):
        return len(self._data)

    len = property(lambda self: len(self._data))

    def __getitem__(self, idx):
        assert idx < len(self._data)
        return self._data[idx]

    def __setitem__(self, idx, el):
        assert idx < len(self._data)
        self._data[idx] = el

    def add(self, el):
        """hittables.nim:59-69: accepts a Hittable or a Sphere."""
        if isinstance(el, Sphere):
            el = Hittable(htSphere, el)
        if not isinstance(el, Hittable):
            raise TypeError("HittablesList.add expects a Sphere or a Hittable")
        self._data.append(el)
        if len(self._data) >= self.size - 1:
            self.size = self.size * 3 // 2

    def items(self):
        return iter(self._data)

    __iter__ = items

    def flatten(self):
        """SoA arrays in list order: dict of float64/uint8 arrays for rtow_set_scene_f64 (Nim `float`
        is binary64; the library rounds what it needs in binary32 itself)."""
        n = len(self._data)
        out = {
            "center_x": np.empty(n, np.float64), "center_y": np.empty(n, np.float64),
            "center_z": np.empty(n, np.float64), "radius": np.empty(n, np.float64),
            "kind": np.empty(n, np.uint8), "albedo": np.empty((n, 3), np.float64),
            "fuzz": np.empty(n, np.float64), "ir": np.empty(n, np.float64),
        }
        for j, h in enumerate(self._data):
            s = h.hSphere
            kind, albedo, fuzz, ir = s.mat.flat()
            out["center_x"][j], out["center_y"][j], out["center_z"][j] = s.center[0], s.center[1], s.center[2]
            out["radius"][j] = s.radius
            out["kind"][j] = kind
            out["albedo"][j] = albedo
            out["fuzz"][j] = fuzz
            out["ir"][j] = ir
        return out


def initHittables(size=8):
    return HittablesList(size)


def hit(h, r, t_min, t_max, rec):
    """hit*(h: var HittablesList, r: Ray, t_min, t_max: float, rec: var HitRecord): bool (hittables.nim:79-88);
    the search over the list runs on the GPU (raytracer.hit)."""
    from . import raytracer
    return raytracer.hit(h, r, t_min, t_max, rec)
