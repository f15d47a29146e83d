"""Sphere and material descriptions with the reference's names (hitrecord.nim:4-49).

The intersection and scatter procs of hitrecord.nim run on the GPU; here are only the value
types a scene is made of.
"""
from . import _lib
from .basetypes import Color, Point, Vec3, dot

mkLambertian, mkMetal, mkDielectric = _lib.LAMBERTIAN, _lib.METAL, _lib.DIELECTRIC


class Lambertian:
    def __init__(self, albedo):
        self.albedo = albedo


class Metal:
    def __init__(self, albedo, fuzz):
        self.albedo = albedo
        self.fuzz = float(fuzz)


class Dielectric:
    def __init__(self, ir):
        self.ir = float(ir)


class Material:
    """Tagged union over the three material objects (hitrecord.nim:18-22)."""

    def __init__(self, kind, payload):
        self.kind = kind
        if kind == mkLambertian:
            self.mLambertian = payload
        elif kind == mkMetal:
            self.mMetal = payload
        elif kind == mkDielectric:
            self.mDielectric = payload
        else:
            raise ValueError(f"unknown MaterialKind {kind}")

    def flat(self):
        """(kind, albedo rgb, fuzz, ir) as plain numbers for the C ABI."""
        if self.kind == mkLambertian:
            a = self.mLambertian.albedo
            return self.kind, (a[0], a[1], a[2]), 0.0, 1.0
        if self.kind == mkMetal:
            a = self.mMetal.albedo
            return self.kind, (a[0], a[1], a[2]), self.mMetal.fuzz, 1.0
        return self.kind, (1.0, 1.0, 1.0), 0.0, self.mDielectric.ir


def initLambertian(a):
    return Lambertian(a)


def initMetal(a, f):
    return Metal(a, f)


def initDielectric(ir):
    return Dielectric(ir)


def initMaterial(m):
    """hitrecord.nim:43-49"""
    if isinstance(m, Lambertian):
        return Material(mkLambertian, m)
    if isinstance(m, Metal):
        return Material(mkMetal, m)
    if isinstance(m, Dielectric):
        return Material(mkDielectric, m)
    raise TypeError(f"initMaterial: {type(m).__name__} is not a material")


class Sphere:
    """hitrecord.nim:11-14"""

    def __init__(self, center, radius, mat=None):
        self.center = center if isinstance(center, Point) else Point(*center)
        self.radius = float(radius)
        self.mat = mat if mat is not None else initMaterial(initLambertian(Color(0, 0, 0)))


class HitRecord:
    """hitrecord.nim:4-9"""

    __slots__ = ("p", "normal", "t", "frontFace", "mat", "index")

    def __init__(self):
        self.p = Point(0, 0, 0)
        self.normal = Vec3(0, 0, 0)
        self.t = 0.0
        self.frontFace = False
        self.mat = None
        self.index = -1   # position of the sphere in its list (not in the reference: a debugging aid)

    def setFaceNormal(self, r, outward_normal):
        """hitrecord.nim:51-53"""
        self.frontFace = dot(r.dir, outward_normal) < 0
        self.normal = outward_normal if self.frontFace else -outward_normal
