"""Host-side value types with the reference's names (basetypes.nim).

These run on the host in binary64, like the Nim originals; they exist so that scenes and cameras
are built exactly as the reference builds them.  Per-ray use of these operations happens on the
GPU (csrc/rtow_path.h), not here.
"""
import math

from . import nimrandom


class Image:
    """basetypes.nim:4-5"""

    def __init__(self, width, height):
        self.width = int(width)
        self.height = int(height)


class Vec3:
    """basetypes.nim:7-8; Color and Point are distinct types over the same storage (:10-11)."""

    __slots__ = ("arr",)

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.arr = [float(x), float(y), float(z)]

    def __getitem__(self, ix):
        return self.arr[ix]

    def __setitem__(self, ix, c):
        self.arr[ix] = float(c)

    def __iter__(self):
        return iter(self.arr)

    def __repr__(self):
        return f"{type(self).__name__}({self.arr[0]!r}, {self.arr[1]!r}, {self.arr[2]!r})"

    def __eq__(self, other):
        return isinstance(other, Vec3) and self.arr == other.arr

    # field accessors (basetypes.nim:34-62)
    x = property(lambda s: s.arr[0])
    y = property(lambda s: s.arr[1])
    z = property(lambda s: s.arr[2])
    r = property(lambda s: s.arr[0])
    g = property(lambda s: s.arr[1])
    b = property(lambda s: s.arr[2])

    def _like(self, a, b, c):
        return type(self)(a, b, c)

    # makeMath (basetypes.nim:68-81): element-wise with a vector, broadcast with a float
    def _bin(self, other, op, swap=False):
        if isinstance(other, Vec3):
            o = other.arr
            return self._like(*(op(o[i], self.arr[i]) if swap else op(self.arr[i], o[i]) for i in range(3)))
        s = float(other)
        return self._like(*(op(s, self.arr[i]) if swap else op(self.arr[i], s) for i in range(3)))

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    def __radd__(self, o):
        return self._bin(o, lambda a, b: a + b, True)

    def __sub__(self, o):
        return self._bin(o, lambda a, b: a - b)

    def __rsub__(self, o):
        return self._bin(o, lambda a, b: a - b, True)

    def __mul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    def __rmul__(self, o):
        return self._bin(o, lambda a, b: a * b, True)

    def __truediv__(self, o):
        return self._bin(o, lambda a, b: a / b)

    def __rtruediv__(self, o):
        return self._bin(o, lambda a, b: a / b, True)

    def __neg__(self):
        return self._like(-self.arr[0], -self.arr[1], -self.arr[2])

    def __pos__(self):
        return self


class Color(Vec3):
    __slots__ = ()


class Point(Vec3):
    __slots__ = ()


def vec3(x, y, z):
    return Vec3(x, y, z)


def color(r, g, b):
    return Color(r, g, b)


def point(x, y, z):
    return Point(x, y, z)


def dot(v, u):
    """basetypes.nim:25-27: accumulated from 0.0 in index order."""
    result = 0.0
    for i in range(3):
        result += v[i] * u[i]
    return result


def cross(v, u):
    """basetypes.nim:29-32"""
    return v._like(v[1] * u[2] - v[2] * u[1], v[2] * u[0] - v[0] * u[2], v[0] * u[1] - v[1] * u[0])


def length_squared(v):
    return v.x * v.x + v.y * v.y + v.z * v.z


def length(v):
    return math.sqrt(length_squared(v))


def normalize(v):
    return v / length(v)


def unitVector(v):
    return normalize(v)


def nearZero(v):
    """basetypes.nim:96-99"""
    epsilon = 1e-8
    return abs(v[0]) < epsilon and abs(v[1]) < epsilon and abs(v[2]) < epsilon


def reflect(v, n):
    """basetypes.nim:101-102"""
    return v - 2 * dot(v, n) * n


def refract(uv, n, etapOverEta):
    """basetypes.nim:104-108"""
    cosTheta = min(dot(-uv, n), 1.0)
    r_out_perp = etapOverEta * (uv + cosTheta * n)
    r_out_parallel = -math.sqrt(abs(1.0 - length_squared(r_out_perp))) * n
    return r_out_perp + r_out_parallel


def randomVec(min=0.0, max=1.0):
    """basetypes.nim:110-112: x, y, z from the global generator, in this order."""
    return Vec3(nimrandom.rand(min, max), nimrandom.rand(min, max), nimrandom.rand(min, max))


def randomInUnitSphere():
    """basetypes.nim:114-118"""
    while True:
        p = randomVec(-1, 1)
        if length_squared(p) >= 1:
            continue
        return p


def randomUnitVector():
    return unitVector(randomInUnitSphere())


def randomInUnitDisk():
    """basetypes.nim:130-134"""
    while True:
        p = vec3(nimrandom.rand(-1.0, 1.0), nimrandom.rand(-1.0, 1.0), 0)
        if length_squared(p) >= 1.0:
            continue
        return p


class Ray:
    """basetypes.nim:13-15"""

    __slots__ = ("orig", "dir")

    def __init__(self, orig, dir):
        self.orig = orig
        self.dir = dir

    def at(self, t):
        """basetypes.nim:170"""
        return Point(*(Vec3(*self.orig) + t * self.dir))


def initRay(origin, direction):
    return Ray(origin, direction)
