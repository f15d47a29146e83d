/* TEST INFRASTRUCTURE (see rtow_oracle.h).  Random number generators of the oracle.
 *
 * (1) Nim std/random restatement.  The reference imports Nim's stdlib generator
 *     (basetypes.nim:1, hitrecord.nim:1, raytracer.nim:1) and calls the global-state
 *     rand()/randomize() at raytracer.nim:41,42,83,84,96,135, basetypes.nim:112,132 and
 *     hitrecord.nim:119.  std/random is not vendored in the reference and its version is not
 *     pinned (no .nimble file); the published algorithm is xoroshiro128+ (Blackman/Vigna 2016
 *     constants 55,14,36) with the fixed default state below.  The restatement is pinned by
 *     the golden values in tests/golden/nim_random.json, which were cross-checked against the
 *     sphere layout of the reference's shipped render.
 *
 * (2) The counter-based generator of the shared deterministic mode: draw k of path
 *     (pixel, sample) is a pure function of (seed, pixel, sample, k).  It replaces the
 *     reference's single sequential stream, which no parallel renderer can reproduce.
 */
#include "rtow_oracle.h"
#include <string.h>

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_nimrand_default(orc_nimrand* r) {
  r->a0 = 0x69B4C98CB8530805ULL;
  r->a1 = 0xD1DD3004688D67CAULL; /* the stdlib literal has 18 hex digits; it truncates to this */
}

uint64_t orc_nimrand_next(orc_nimrand* r) {
  const uint64_t s0 = r->a0;
  uint64_t s1 = r->a1;
  const uint64_t result = s0 + s1;
  s1 ^= s0;
  r->a0 = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
  r->a1 = rotl64(s1, 36);
  return result;
}

/* skipRandomNumbers: the 2^64-step jump Nim >= 1.6 performs inside initRand */
static void nimrand_jump(orc_nimrand* r) {
  static const uint64_t helper[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t s0 = 0, s1 = 0;
  for (int i = 0; i < 2; ++i)
    for (int b = 0; b < 64; ++b) {
      if (helper[i] & (1ULL << b)) { s0 ^= r->a0; s1 ^= r->a1; }
      (void)orc_nimrand_next(r);
    }
  r->a0 = s0;
  r->a1 = s1;
}

void orc_nimrand_init(orc_nimrand* r, int64_t seed, int with_jump) {
  r->a0 = (uint64_t)seed >> 16;
  r->a1 = (uint64_t)seed & 0xffff;
  if (with_jump) nimrand_jump(r);
  (void)orc_nimrand_next(r);
}

double orc_nimrand_rand(orc_nimrand* r, double max) {
  const uint64_t x = orc_nimrand_next(r);
  const uint64_t bits = (0x3FFULL << 52) | (x >> 12);
  double d;
  memcpy(&d, &bits, sizeof d);
  return (d - 1.0) * max;
}

/* ------------------------------------------------------------------ counter generator */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

void orc_path_key(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t key[2]) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  key[0] = mix32(mix32(pixel ^ s0) + sample);
  key[1] = mix32(mix32(sample ^ s1 ^ 0xA511E9B3U) + pixel);
}

uint32_t orc_draw_u32(const uint32_t key[2], uint32_t k) {
  uint32_t x = key[0] + k * 0x9E3779B9U;
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x ^= key[1]; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

double orc_draw_unit(const uint32_t key[2], uint32_t k) {
  return (double)(orc_draw_u32(key, k) >> 8) * (1.0 / 16777216.0);
}
