"""The scene container with the reference's interface (hittables.nim:3-73).

The reference grows a raw buffer by hand; a Python list gives the same observable behaviour
(append order = list index, which decides ties in the closest-hit scan, hittables.nim:84-88).
`flatten` produces the structure-of-arrays form the C ABI uploads.
"""
import numpy as np

from .hitrecord import Sphere

htSphere = 0


class Hittable:
    """hittables.nim:4-8: a variant with a single member today.  `kind` is kept as given: a value other than
    htSphere travels to the library with the list and is refused there (rtow_set_hittables), as the reference's
    `case h.kind` (hittables.nim:75-77) has no branch for it."""

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
            "hittable_kind": np.empty(n, np.uint8),    # Hittable.kind per entry (hittables.nim:6)
        }
        for j, h in enumerate(self._data):
            out["hittable_kind"][j] = h.kind
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
    """The reference's three overloads of hit*, told apart by the type of `h` as Nim's overload resolution does:
      hit*(h: var HittablesList, r, t_min, t_max, rec)  hittables.nim:79-88  closest hit over the list
      hit*(h: Hittable, r, t_min, t_max, rec)           hittables.nim:75-77  case h.kind of htSphere: h.hSphere.hit
      hit*(s: Sphere, r, t_min, t_max, rec)             hitrecord.nim:55-79  one sphere
    All three run on the GPU (raytracer.hit); the one-object forms are a query against a one-entry list."""
    from . import raytracer
    if isinstance(h, Hittable):
        if h.kind != htSphere:   # the reference's case statement has no other branch
            raise ValueError(f"hit: Hittable of kind {h.kind}; the only HittableKind is htSphere ({htSphere})")
        h = h.hSphere
    if isinstance(h, Sphere):
        one = HittablesList()
        one.add(h)
        return raytracer.hit(one, r, t_min, t_max, rec)
    return raytracer.hit(h, r, t_min, t_max, rec)
