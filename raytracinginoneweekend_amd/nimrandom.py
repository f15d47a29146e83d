"""Nim `std/random` as the reference uses it for scene construction.

`randomScene()` runs before `randomize(0xE7)` (raytracer.nim:122 vs :135), so the scene comes
from the un-seeded global generator: xoroshiro128+ from a fixed state.  Only the calls the
reference makes are provided: rand(max: float), rand(a..b) and randomize(seed).
"""
import struct

_MASK = (1 << 64) - 1


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & _MASK


class Rand:
    """State of one generator; `Rand()` is the un-seeded global state."""

    def __init__(self, a0=0x69B4C98CB8530805, a1=0xD1DD3004688D67CA):
        self.a0 = a0
        self.a1 = a1
        self.draws = 0

    def next(self):
        s0, s1 = self.a0, self.a1
        result = (s0 + s1) & _MASK
        s1 ^= s0
        self.a0 = _rotl(s0, 55) ^ s1 ^ ((s1 << 14) & _MASK)
        self.a1 = _rotl(s1, 36)
        self.draws += 1
        return result

    def rand(self, max_or_lo, hi=None):
        """rand(max) -> [0, max);  rand(lo, hi) = rand(lo .. hi) = rand(hi - lo) + lo."""
        if hi is not None:
            return self.rand(hi - max_or_lo) + max_or_lo
        bits = (0x3FF << 52) | (self.next() >> 12)
        return (struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0) * max_or_lo


def initRand(seed, with_jump=False):
    """initRand / randomize(seed).  Nim >= 1.6 additionally jumps the fresh state 2^64 steps."""
    r = Rand((seed & _MASK) >> 16, seed & 0xFFFF)
    if with_jump:
        s0 = s1 = 0
        for helper in (0xBEAC0467EBA5FACB, 0xD86B048B86AA9922):
            for b in range(64):
                if helper & (1 << b):
                    s0 ^= r.a0
                    s1 ^= r.a1
                r.next()
        r.a0, r.a1 = s0, s1
    r.next()
    r.draws = 0
    return r


# the module-level generator the reference's rand()/randomize() act on
_global = Rand()


def globalState():
    return _global


def resetGlobal():
    global _global
    _global = Rand()
    return _global


def rand(max_or_lo, hi=None):
    return _global.rand(max_or_lo, hi)


def randomize(seed, with_jump=False):
    global _global
    _global = initRand(seed, with_jump)
