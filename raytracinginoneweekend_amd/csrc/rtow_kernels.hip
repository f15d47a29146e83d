// gfx950 kernels of the renderer.
//
//  rtow_trace_kernel   persistent waves; every lane carries one path (pixel, sample) through an
//                      iterative bounce loop and is refilled with a new path the moment its
//                      path ends, so the sphere scan always runs with full wavefronts.  Work is
//                      handed out in chunks (runs of consecutive paths, pixel-major) from a global
//                      counter, in sizes that shrink as the frame runs out.  Radiance is summed per lane in
//                      64-bit fixed point; a lane that moves on to another pixel adds its sums
//                      (through a small per-wave cache in LDS) to the frame with integer atomics,
//                      which makes the image independent of scheduling, chunking and GPU count.
//  rtow_resolve_kernel mean + sqrt gamma + clamp + 8-bit quantisation (raytracer.nim:22-28).
//
// The sphere scan (hittables.nim:79-88 over hitrecord.nim:55-63) keeps the ray in VGPRs and
// streams the sphere table through SGPRs with wave-uniform scalar loads, four spheres per load,
// two spheres per packed FP32 instruction, and a rarely taken branch for discriminant >= 0.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "rtow_kernels.h"

namespace rtow {

#ifdef RTOW_PROFILE_COUNTERS
// debug build only: [0] wave-level group slow-path entries, [1] wave-level candidate evaluations,
// [2] lane-level candidate evaluations, [3] wave iterations, [4] lane iterations (alive lanes)
__device__ unsigned long long g_prof[16];

// wave-local accumulation (values are wave-uniform), flushed once when the wave retires
#define RTOW_PROF_DECL unsigned long long prof_[16] = {0}
#define RTOW_PROF_PARAM , unsigned long long (&prof_)[16]
#define RTOW_PROF_ARG , prof_
#define RTOW_COUNT(i, v) do { prof_[i] += (unsigned long long)(v); } while (0)
#define RTOW_PROF_FLUSH() do { if ((threadIdx.x & 63u) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_prof[i_], prof_[i_]); } while (0)
#define RTOW_CLOCK() __builtin_amdgcn_s_memtime()
#elif defined(RTOW_PROFILE_TIMELINE)
// timeline build: per-wave time stamps only (plain stores to distinct addresses, no shared counters)
#define RTOW_PROF_DECL
#define RTOW_PROF_PARAM
#define RTOW_PROF_ARG
#define RTOW_COUNT(i, v) do { } while (0)
#define RTOW_PROF_FLUSH() do { } while (0)
#define RTOW_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define RTOW_PROF_DECL
#define RTOW_PROF_PARAM
#define RTOW_PROF_ARG
#define RTOW_COUNT(i, v) do { } while (0)
#define RTOW_PROF_FLUSH() do { } while (0)
#define RTOW_CLOCK() 0ull
#endif

#ifdef RTOW_PROFILE_TIMELINE
// per wave, 12 slots: realtime stamps (100 MHz) at start [0], when the work counter ran dry [1], at
// the end [2]; live lanes [3] and iterations [4] at that point, iterations at the end [5]; core-clock
// stamps of the last iteration [6..9]; realtime at loop exit [10] and at the end of that iteration [11]
__device__ unsigned long long g_wave_times[12 * 16384];
#endif

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
// constant address space: wave-uniform loads from it become s_load_dwordx16 into SGPRs
typedef const f16v __attribute__((address_space(4)))* geom_ptr;
typedef float f4v __attribute__((ext_vector_type(4)));

// per-ray operands of the packed filter, every component duplicated into both halves of a pair
struct ScanRay {
  f2v ux, uy, uz, nou;     // unit direction, -(o.u)
  f2v ox2, oy2, oz2;       // 2*o
  float threshold;         // |o|^2 - margin
};

static __device__ __forceinline__ ScanRay scan_ray(f3 o, f3 u, float cmax) {
  ScanRay r;
  const FilterRay f = filter_ray(o, u, cmax);
  r.ux = f2v{u.x, u.x}; r.uy = f2v{u.y, u.y}; r.uz = f2v{u.z, u.z};
  r.nou = f2v{f.nou, f.nou};
  r.ox2 = f2v{f.ox2, f.ox2}; r.oy2 = f2v{f.oy2, f.oy2}; r.oz2 = f2v{f.oz2, f.oz2};
  r.threshold = f.threshold;
  return r;
}

static __device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Closest hit over the regular (binary32) table: a cheap packed filter over every sphere, and the
// exact Sphere.hit arithmetic (hitrecord.nim:56-63) only for the spheres the filter lets through.
//
// Filter.  With n = c.u - o.u and k = |c|^2 - r^2 the discriminant is
//     disc = n^2 - (k + |o|^2 - 2 o.c),
// so  disc >= -margin  <=>  n^2 + (2o).c - k >= |o|^2 - margin.   The left side costs eight
// packed instructions per PAIR of spheres (no per-sphere subtraction of the centre), the right
// side is one number per ray.  Rounding makes the filter value differ from the exact
// discriminant by a few ulps of (|o| + |c|)^2, which `margin` (ScanRay::threshold) covers; a sphere
// that passes is then tested exactly, so the filter only ever adds work, never hits.
//
// The table is stored in groups of four spheres, 16 floats per group, as two packed pairs and k:
//   cx0 cx1  cy0 cy1  cz0 cz1 | cx2 cx3  cy2 cy3  cz2 cz3 | k0 k1 k2 k3
// Scalar loads bring a group into SGPRs, two groups ahead of its use: the first 12 floats when -k
// comes from LDS (scalar registers are the scarcest resource of this kernel), all 16 otherwise.  Spheres handled
// elsewhere and the padding carry k = +inf, so they never pass.  r^2 for the exact test comes from
// a separate array, fetched only when needed.
static __device__ __forceinline__ f2v pair_filter(const f2v cx, const f2v cy, const f2v cz, const f2v nk,
                                                  const ScanRay& r) {
  f2v n = __builtin_elementwise_fma(cx, r.ux, r.nou);
  n = __builtin_elementwise_fma(cy, r.uy, n);
  n = __builtin_elementwise_fma(cz, r.uz, n);
  f2v t = __builtin_elementwise_fma(cx, r.ox2, nk);   // nk = -k, a VGPR pair (see K_LDS)
  t = __builtin_elementwise_fma(cy, r.oy2, t);
  t = __builtin_elementwise_fma(cz, r.oz2, t);
  return __builtin_elementwise_fma(n, n, t);
}

// Spheres that pass the filter are not tested on the spot (only a few lanes of the wave would be
// busy).  Survivors go onto ONE list per wave in LDS as (lane, list index) words, and the list is
// worked off by all 64 lanes together - lane L takes entries L, L + 64, ... whoever they belong to:
// it fetches the owner's ray across the wave (ds_bpermute), runs the exact test and offers the root to
// the owner's closest-hit word, a sortable 64-bit key in LDS updated with ds_min_u64.  A wave has
// about 80 survivors per scan, unevenly spread over its lanes: per-lane queues took as many rounds as
// the longest of them (3.9 on average), the shared list takes total / 64 (1.5 on average).
// The closest-hit rule (smaller t, then larger index) does not depend on the order of the offers.
typedef unsigned long long __attribute__((address_space(3)))* lds_u64_ptr;
constexpr uint32_t kListEntries = 512;   // per wave; drained when fewer than 256 (one group, all lanes) are free
constexpr uint32_t kOwnerShift = 26;     // entry = lane << 26 | list index
static_assert(kCandWords == (int)kListEntries + 128, "per-wave list + 64 keys");
static_assert(kMaxSpheres + 16 <= (1 << kOwnerShift), "list index must fit beside the lane number");

struct CandList {
  uint32_t* list;     // LDS: kListEntries words of this wave
  lds_u64_ptr best;   // LDS: 64 words of this wave, lane L's closest hit so far (hit_key)
  uint32_t n;         // wave-uniform: entries in the list
  uint32_t lane;
};

// smaller t first, then the larger list index; id = -1 (nothing hit) loses every tie
static __device__ __forceinline__ unsigned long long hit_key(float t, int id) {
  return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)(0x7fffffff - id);
}

static __device__ __forceinline__ void lds_order() {
  // one wave's LDS requests are served in order; the compiler only has to keep them in order too
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform control flow; `pass` differs per lane
static __device__ __forceinline__ void cand_push(CandList& c, bool pass, int j) {
  // Usually one of a group's four spheres has survivors; the wave-uniform branch skips the other three
  // pushes (instructions issued with an empty execution mask still take their issue slots).
  const uint64_t m = __ballot(pass);
  if (m != 0) {
    if (pass) c.list[c.n + lanes_below(m)] = (c.lane << kOwnerShift) | (uint32_t)j;
    c.n += (uint32_t)__popcll(m);
  }
}

static __device__ __forceinline__ float from_lane(int byte_index, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_index, __float_as_int(v)));
}

// Must run with all 64 lanes active (rays are read from their owners' registers).
template <typename ExactPtr>
static __device__ __forceinline__ void cand_drain(CandList& c, ExactPtr exact, f3 o, f3 u, float tmin,
                                                  int prev RTOW_PROF_PARAM) {
  lds_order();
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.n);   // a scalar loop bound
  for (uint32_t base = 0; base < n; base += 64u) {
    RTOW_COUNT(10, 1);
    const uint32_t k = base + c.lane;
    const bool has = k < n;
    const uint32_t entry = c.list[k];   // lanes past the end read into the key words behind the list: unused
    const int owner = (int)((entry >> kOwnerShift) << 2);
    const int j = (int)(entry & ((1u << kOwnerShift) - 1u));
    const f3 oo = {from_lane(owner, o.x), from_lane(owner, o.y), from_lane(owner, o.z)};
    const f3 uu = {from_lane(owner, u.x), from_lane(owner, u.y), from_lane(owner, u.z)};
    const float tm = from_lane(owner, tmin);
    const int pv = __builtin_amdgcn_ds_bpermute(owner, prev);
    if (has) {
      const f4v sp = exact[j];   // (cx, cy, cz, r^2)
      const float ocx = oo.x - sp.x, ocy = oo.y - sp.y, ocz = oo.z - sp.z;
      float b = ocx * uu.x; b = fmaf(ocy, uu.y, b); b = fmaf(ocz, uu.z, b);
      float m = fmaf(-ocx, ocx, sp.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
      const float disc = fmaf(b, b, m);
      float t;
      const bool ok = regular_candidate(b, disc, tm, j == pv, t);
      if (disc >= 0.0f && ok)
        __hip_atomic_fetch_min(c.best + (entry >> kOwnerShift), hit_key(t, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  c.n = 0;
  lds_order();
}

struct GroupRec {   // centres of four spheres as packed pairs
  f8v a;            // cx0 cx1 cy0 cy1 cz0 cz1 cx2 cx3
  f4v b;            // cy2 cy3 cz2 cz3
};

template <bool K_LDS>
static __device__ __forceinline__ GroupRec load_group(geom_ptr geom, int g, f4v& nk) {
  GroupRec r;
  if (K_LDS) {
    typedef const f8v __attribute__((address_space(4)))* p8;
    typedef const f4v __attribute__((address_space(4)))* p4;
    r.a = *(p8)(geom + g);
    r.b = *((p4)(geom + g) + 2);
  } else {
    const f16v v = geom[g];
    r.a = v.lo;
    r.b = f4v{v.s8, v.s9, v.sa, v.sb};
    nk = f4v{-v.sc, -v.sd, -v.se, -v.sf};
  }
  return r;
}

template <typename ExactPtr>
static __device__ __forceinline__ void group_test(const GroupRec grp, const f4v nk, int j, ExactPtr exact,
                                                  CandList& cand, const ScanRay& r, f3 o, f3 u, float tmin,
                                                  int prev RTOW_PROF_PARAM) {
  const f2v f0 = pair_filter(grp.a.s01, grp.a.s23, grp.a.s45, nk.xy, r);
  const f2v f1 = pair_filter(grp.a.s67, grp.b.xy, grp.b.zw, nk.zw, r);
  const float mx = fmaxf(fmaxf(f0.x, f0.y), fmaxf(f1.x, f1.y));
  if (__ballot(mx >= r.threshold) != 0) {   // wave-uniform: the list's fill count stays a scalar
    RTOW_COUNT(9, 1);
    RTOW_COUNT(0, mx >= r.threshold ? 1 : 0);
    cand_push(cand, f0.x >= r.threshold, j);
    cand_push(cand, f0.y >= r.threshold, j + 1);
    cand_push(cand, f1.x >= r.threshold, j + 2);
    cand_push(cand, f1.y >= r.threshold, j + 3);
    if (cand.n > kListEntries - 256u) cand_drain(cand, exact, o, u, tmin, prev RTOW_PROF_ARG);
  }
}

// n_groups is even and the table holds one more (all-padding) group after the last one, so the
// loop can always request two groups ahead without a bounds test.
//
// K_LDS: -k of every sphere sits in LDS (copied once per workgroup) and reaches the filter as a
// VGPR pair through a broadcast ds_read_b128 issued one group ahead.  That lets the second chain
// start as fma(cx, 2o.x, -k): a VALU instruction can read only one SGPR operand, so with k in an
// SGPR the subtraction would cost a packed instruction of its own.  Tables too large for LDS use
// the SGPR copy of k (moved to VGPRs first; same arithmetic, two more instructions per group).
typedef const f4v __attribute__((address_space(3)))* nk_lds_ptr;

// Called by all 64 lanes of the wave; lanes without a path pass live = false and offer nothing.
template <bool K_LDS, typename ExactPtr>
static __device__ __forceinline__ void scan_regular(geom_ptr geom, ExactPtr exact, const f4v* nk_lds,
                                                    CandList& cand, int n_groups, float cmax, bool live, f3 o, f3 u,
                                                    float tmin, int prev, float& best, int& id RTOW_PROF_PARAM) {
  if (n_groups <= 0) return;
  ScanRay r = scan_ray(o, u, cmax);
  if (!live) r.threshold = INFINITY;
  cand.best[cand.lane] = hit_key(best, id);   // what the binary64 spheres left; read back after the last drain
  // The LDS address of -k lives in a VGPR that advances by two groups per trip; the reads then use
  // immediate offsets.  (Left to itself the compiler keeps the address in an SGPR and copies it to
  // a VGPR for every read: one more VALU instruction per group.)
  uint32_t nk_addr = K_LDS ? (uint32_t)(uintptr_t)(nk_lds_ptr)nk_lds : 0u;
  asm volatile("" : "+v"(nk_addr));
  nk_lds_ptr nk = (nk_lds_ptr)(uintptr_t)nk_addr;
  f4v ka = K_LDS ? nk[0] : f4v{0, 0, 0, 0}, kb = ka;
  GroupRec ga = load_group<K_LDS>(geom, 0, ka);
  for (int g = 0; g < n_groups; g += 2) {
    const GroupRec gb = load_group<K_LDS>(geom, g + 1, kb);
    if (K_LDS) kb = nk[1];
    group_test(ga, ka, 4 * g, exact, cand, r, o, u, tmin, prev RTOW_PROF_ARG);
    ga = load_group<K_LDS>(geom, g + 2, ka);
    if (K_LDS) { ka = nk[2]; nk += 2; }
    group_test(gb, kb, 4 * g + 4, exact, cand, r, o, u, tmin, prev RTOW_PROF_ARG);
  }
  cand_drain(cand, exact, o, u, tmin, prev RTOW_PROF_ARG);
  const unsigned long long key = cand.best[cand.lane];
  best = __uint_as_float((uint32_t)(key >> 32));
  id = 0x7fffffff - (int)(uint32_t)key;
}

// The same closest-hit query for a wave that has only a few paths left (the end of a frame, when no
// work is left to refill the idle lanes): instead of every lane walking the whole table for its own
// ray, the rays are taken one at a time and the TABLE is spread over the 64 lanes - lane L screens
// spheres L, L + 64, ... for the ray of the current owner, read from the records in LDS.  The screen
// is the same arithmetic (filter_value is pair_filter one sphere at a time), survivors go to the same
// list with the owner's lane number, and the exact tests are shared as always, so the result is the
// same hit; a wave with m paths pays about m * 100 instructions instead of 2 000.
// Needs -k and the exact-test records in LDS (tables up to kMaxSpheresExactInLds entries).
constexpr uint32_t kSparseLanes = 12;

template <typename ExactPtr>
static __device__ __forceinline__ void scan_sparse(ExactPtr exact, const f4v* nk_lds, CandList& cand, int n_pad,
                                                   float cmax, bool live, f3 o, f3 u, float tmin, int prev,
                                                   float& best, int& id RTOW_PROF_PARAM) {
  typedef const float __attribute__((address_space(3)))* lds_f32_ptr;
  const lds_f32_ptr nk = (lds_f32_ptr)(const float*)nk_lds;
  cand.best[cand.lane] = hit_key(best, id);
  uint64_t todo = __ballot(live);
  while (todo != 0) {
    const int owner = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int ob = owner << 2;
    const f3 oo = {from_lane(ob, o.x), from_lane(ob, o.y), from_lane(ob, o.z)};
    const f3 uu = {from_lane(ob, u.x), from_lane(ob, u.y), from_lane(ob, u.z)};
    const FilterRay fr = filter_ray(oo, uu, cmax);
    for (int j0 = 0; j0 < n_pad; j0 += 64) {
      const int j = j0 + (int)cand.lane;
      bool pass = false;
      if (j < n_pad) {
        const f4v sp = exact[j];
        pass = filter_value(sp.x, sp.y, sp.z, -nk[j], uu, fr) >= fr.threshold;   // padding and binary64 spheres: k = +inf
      }
      const uint64_t m = __ballot(pass);
      if (m != 0) {
        if (pass) cand.list[cand.n + lanes_below(m)] = ((uint32_t)owner << kOwnerShift) | (uint32_t)j;
        cand.n += (uint32_t)__popcll(m);
        if (cand.n > kListEntries - 64u) cand_drain(cand, exact, o, u, tmin, prev RTOW_PROF_ARG);
      }
    }
  }
  cand_drain(cand, exact, o, u, tmin, prev RTOW_PROF_ARG);
  const unsigned long long key = cand.best[cand.lane];
  best = __uint_as_float((uint32_t)(key >> 32));
  id = 0x7fffffff - (int)(uint32_t)key;
}

static __device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// A lane keeps the fixed-point radiance sums of the pixel it is working on in registers (about eight
// paths in a row belong to the same pixel at spp 500) and hands them over when it moves on to another
// pixel.  The lanes of a wave change pixel a few at a time, so the sums of a pixel are collected in a
// small per-wave cache in LDS first - four direct-mapped slots of three 64-bit words, filled with LDS
// atomics - and a slot goes to the frame (three 64-bit global atomics) when another pixel claims it
// or the wave ends: one set of global atomics per pixel and wave instead of one per lane.  Integer
// sums do not depend on the order they arrive in.  (A shuffle reduction over the lanes of a pixel does
// the same at ~100 VALU instructions per round; the LDS atomics cost the VALU one instruction each.)
constexpr uint32_t kSumSlots = 4;
constexpr uint32_t kSumCacheDwords = 32;   // per wave: kSumSlots * 3 words of 8 bytes, padded to 128 bytes
constexpr uint32_t kNoPixel = 0xffffffffu;

struct SumCache {
  lds_u64_ptr words;             // this wave's kSumSlots * 3 words
  uint32_t tags;                 // lane s < kSumSlots: the pixel slot s holds (one VGPR; scalar registers are scarce here)
};

static __device__ __forceinline__ void sum_cache_evict(unsigned long long* accum, const SumCache& c, uint32_t slot,
                                                       uint32_t pix, uint32_t lane) {
  // LDS serves one wave's requests in order: the adds of earlier instructions are in the words read here
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 3u) {
    const unsigned long long v = c.words[3u * slot + lane];
    c.words[3u * slot + lane] = 0ull;
    atomicAdd(&accum[3ull * pix + lane], v);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Must be called by all 64 lanes together; lanes with `want` set hand over (accpix, acc_*).
static __device__ __forceinline__ void flush_sums(unsigned long long* accum, SumCache& c, bool want, uint32_t accpix,
                                                  unsigned long long acc_r, unsigned long long acc_g,
                                                  unsigned long long acc_b, uint32_t lane) {
  uint64_t todo = __ballot(want);
  while (todo != 0) {
    const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
    const uint32_t pix = (uint32_t)__builtin_amdgcn_readlane((int)accpix, (int)leader);
    const uint32_t slot = pix & (kSumSlots - 1u);
    const uint32_t held = (uint32_t)__builtin_amdgcn_readlane((int)c.tags, (int)slot);
    if (held != pix) {
      if (held != kNoPixel) sum_cache_evict(accum, c, slot, held, lane);
      c.tags = lane == slot ? pix : c.tags;
    }
    const bool mine = want && accpix == pix;
    if (mine) {
      __hip_atomic_fetch_add(c.words + 3u * slot + 0u, acc_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(c.words + 3u * slot + 1u, acc_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(c.words + 3u * slot + 2u, acc_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    todo &= ~__ballot(mine);
  }
}

static __device__ __forceinline__ void sum_cache_drain(unsigned long long* accum, SumCache& c, uint32_t lane) {
  for (uint32_t slot = 0; slot < kSumSlots; ++slot) {
    const uint32_t held = (uint32_t)__builtin_amdgcn_readlane((int)c.tags, (int)slot);
    if (held != kNoPixel) sum_cache_evict(accum, c, slot, held, lane);
  }
  c.tags = kNoPixel;
}

// K_LDS: -k per sphere in LDS (tables up to kMaxSpheresKInLds entries).  X_LDS: the records of the
// exact test (16 bytes per sphere) in LDS as well (tables up to kMaxSpheresExactInLds entries): a
// round of exact tests then waits for LDS, not for L2.
template <bool K_LDS, bool X_LDS>
__global__ void __launch_bounds__(kTraceThreads, 5) rtow_trace_kernel(const TraceParams P) {
  // LDS: candidate lists + closest-hit keys [4 waves][kCandWords], path pools [4 waves][kPoolFields][64], sum caches
  // [4 waves][kSumCacheDwords], then
  // (K_LDS) -k per sphere, then (X_LDS) the exact-test records
  extern __shared__ f4v lds_raw[];
  uint32_t* const stacks = (uint32_t*)lds_raw;
  constexpr uint32_t kPerWaveDwords = kCandWords + kPoolFields * 64 + kSumCacheDwords;
  const f4v* const nk_lds = lds_raw + (kTraceThreads / 64) * kPerWaveDwords / 4;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t* const pool = stacks + (kTraceThreads / 64) * kCandWords + wave * (kPoolFields * 64);
  const geom_ptr geom = (geom_ptr)(uintptr_t)P.geom;
  CandList cand;
  cand.list = stacks + wave * kCandWords;
  cand.best = (lds_u64_ptr)(unsigned long long*)(cand.list + kListEntries);
  cand.n = 0;
  cand.lane = lane;
  SumCache sums;
  {
    uint32_t* const mine = stacks + (kTraceThreads / 64) * (kCandWords + kPoolFields * 64) + wave * kSumCacheDwords;
    if (lane < kSumCacheDwords) mine[lane] = 0u;   // read back only by this wave, in order
    sums.words = (lds_u64_ptr)(unsigned long long*)mine;
    sums.tags = kNoPixel;
  }
  if (K_LDS) {
    const int n4 = P.n_geom_pad / 4 + 1;
    f4v* const dst = lds_raw + (kTraceThreads / 64) * kPerWaveDwords / 4;
    for (int i = threadIdx.x; i < n4; i += kTraceThreads) dst[i] = ((const f4v*)P.geom_nk)[i];
    if (X_LDS) {
      f4v* const xdst = dst + n4;
      for (int i = threadIdx.x; i < P.n_geom_pad; i += kTraceThreads) xdst[i] = ((const f4v*)P.geom_exact)[i];
    }
    __syncthreads();
  }
  typedef const f4v __attribute__((address_space(3)))* exact_lds_ptr;
  const exact_lds_ptr exact_lds = (exact_lds_ptr)(nk_lds + (P.n_geom_pad / 4 + 1));

  // wave-uniform: the chunk being handed out.  A chunk is a run of consecutive paths in
  // pixel-major order (all samples of a pixel, then the next pixel of the row list).
  uint32_t e_next = 0, e_end = 0;     // offsets inside the chunk
  unsigned long long seen = 0;        // first path of this wave's previous chunk
  uint32_t pool_n = 0, pool_head = 0; // camera rays generated into the pool / already handed to lanes
  uint32_t p_base = 0, s_base = 0;    // pixel (index in the row list's image) and sample of the chunk's first path
  uint32_t r_base = 0, i_base = 0;    // row (in the row list) and column of that pixel
  bool exhausted = false;

  // per lane: the path in flight
  bool alive = false;
  f3 o = {0, 0, 0}, d = {0, 0, 1};
  float tr = 0, tg = 0, tb = 0;
  Stream g = {0, 0};
  int depth = 0, prev = -1;
  bool first_sample = false;
  // per lane: radiance sums of the pixel it is working on
  uint32_t accpix = 0xffffffffu;
  unsigned long long acc_r = 0, acc_g = 0, acc_b = 0;
  uint32_t nseg = 0;
  float my_paths = 0.0f;              // paths this wave has been handed so far
  const float inv_waves = 1.0f / (float)P.n_waves;
  RTOW_PROF_DECL;
#ifdef RTOW_PROFILE_TIMELINE
  const uint32_t wave_id = blockIdx.x * (kTraceThreads / 64) + (threadIdx.x >> 6);
  if (lane == 0 && wave_id < 16384) g_wave_times[12 * wave_id] = __builtin_amdgcn_s_memrealtime();
  unsigned long long iters_ = 0;
#endif

  for (;;) {
    const unsigned long long t_refill = RTOW_CLOCK();
    // ---- refill idle lanes
    for (;;) {
      const uint64_t need = __ballot(!alive);
      if (need == 0) break;
      if (pool_head >= pool_n) {
       if (e_next >= e_end) {
        if (exhausted) break;
        // guided self-scheduling: a quarter of an even share of what was left when this wave
        // last looked, clamped to [chunk_min, chunk_max] and rounded down to whole wavefronts
        uint32_t want = P.chunk_max;
        const unsigned long long left_seen = P.total_paths - seen;
        if (left_seen < 4ull * P.n_waves * P.chunk_max) {
          want = ((uint32_t)left_seen / (4u * P.n_waves)) & ~63u;
          want = want < P.chunk_min ? P.chunk_min : want;
        }
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(P.path_counter, (unsigned long long)want);
        // (the builtin returns int: widen through uint32_t, or a low word >= 2^31 sign-extends over the high word)
        first = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(first >> 32)) << 32) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first);
        if (first >= P.total_paths) {
#ifdef RTOW_PROFILE_TIMELINE
          {
            const unsigned long long live_ = __popcll(__ballot(alive));
            if (lane == 0 && wave_id < 16384) {
              g_wave_times[12 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
              g_wave_times[12 * wave_id + 3] = live_;
              g_wave_times[12 * wave_id + 4] = iters_;
            }
          }
#endif
          exhausted = true;
          break;
        }
        seen = first;
        {
          // The SIMD's arbiter serves its resident waves by priority, then age, and age never
          // changes for persistent waves: left alone, the oldest wave of a SIMD runs ten times as
          // many iterations as the youngest, which then sits on stale work when the frame runs out
          // and keeps the grid alive for milliseconds.  Each wave therefore compares what it has
          // consumed with an even share of what has been handed out and raises its priority when it
          // is behind, lowers it when it is ahead.
          const float ahead = my_paths - (float)first * inv_waves;
          if (ahead < -512.0f) __builtin_amdgcn_s_setprio(3);
          else if (ahead < 0.0f) __builtin_amdgcn_s_setprio(2);
          else if (ahead < 512.0f) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
          my_paths += (float)want;
        }
        const unsigned long long left = P.total_paths - first;
        // Pixels are handed out from the LAST pixel of the row list backwards, so a frame ends on its
        // top rows.  The waves retire only when their longest path has ended, and in the scenes this
        // renderer is used on the top of the frame is sky (one segment per path) while the bottom is
        // ground and spheres (paths of up to max_depth segments): finishing on cheap paths cuts the
        // time the grid spends draining from milliseconds to one iteration.
        const uint32_t p_seq = (uint32_t)(first / (uint32_t)P.spp);
        s_base = (uint32_t)(first - (unsigned long long)p_seq * (uint32_t)P.spp);
        p_base = (uint32_t)P.n_rows * (uint32_t)P.W - 1u - p_seq;
        r_base = p_base / (uint32_t)P.W;
        i_base = p_base - r_base * (uint32_t)P.W;
        e_next = 0;
        e_end = left < want ? (uint32_t)left : want;
       }
        // ---- generate: the next (up to) 64 paths of the chunk, one per lane, ALL lanes at work
        // The camera (19 scalars) is read from the kernel arguments here, every time: kept in scalar
        // registers across the sphere scan it pushes other values out into vector-register lanes.
        typedef const float __attribute__((address_space(4)))* karg_ptr;
        karg_ptr cf = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();   // TraceParams starts with cam: 19 floats
        asm volatile("" : "+s"(cf));
        static_assert(offsetof(TraceParams, cam) == 0 && sizeof(CameraParams) == 19 * sizeof(float), "camera layout");
        CameraParams cam;
        cam.origin = {cf[0], cf[1], cf[2]}; cam.llc = {cf[3], cf[4], cf[5]}; cam.hor = {cf[6], cf[7], cf[8]};
        cam.ver = {cf[9], cf[10], cf[11]}; cam.u = {cf[12], cf[13], cf[14]}; cam.v = {cf[15], cf[16], cf[17]};
        cam.lens_radius = cf[18];
        const uint32_t n_gen = min(64u, e_end - e_next);
        const bool gen = lane < n_gen;
        // (pixel, sample) of lane's path: e counts samples from the first sample of the chunk's
        // first pixel.  With spp >= chunk_max it overflows spp at most once.
        const uint32_t e = s_base + e_next + (gen ? lane : 0u);
        uint32_t dp, sample;
        if ((uint32_t)P.spp >= P.chunk_max) {
          dp = e >= (uint32_t)P.spp ? 1u : 0u;
          sample = e - (dp ? (uint32_t)P.spp : 0u);
        } else {
          // e < 2 * chunk_max + spp and spp < chunk_max here: exact (2^32 / 1 does not fit inv_spp)
          dp = P.spp == 1 ? e : __umulhi(e, P.inv_spp);
          sample = e - dp * (uint32_t)P.spp;
        }
        if (gen) {
          uint32_t r = r_base;
          int ci = (int)i_base - (int)dp;
          while (ci < 0) { ci += P.W; --r; }
          const int y = P.rows ? P.rows[r] : (int)r;
          Stream g2 = stream_begin(P.seed_lo, P.seed_hi, (uint32_t)y * (uint32_t)P.W + (uint32_t)ci, sample + P.sample_base);
          f3 o2, d2;
          camera_ray(cam, ci, P.H - 1 - y, P.W, P.H, g2, o2, d2);
          pool[0 * 64 + lane] = __float_as_uint(o2.x); pool[1 * 64 + lane] = __float_as_uint(o2.y);
          pool[2 * 64 + lane] = __float_as_uint(o2.z); pool[3 * 64 + lane] = __float_as_uint(d2.x);
          pool[4 * 64 + lane] = __float_as_uint(d2.y); pool[5 * 64 + lane] = __float_as_uint(d2.z);
          pool[6 * 64 + lane] = g2.x0; pool[7 * 64 + lane] = g2.key1;
          // local pixel index (< 2^31 / 3); top bit: sample 0 of the pixel (it records the hit-id plane)
          pool[8 * 64 + lane] = (p_base - dp) | ((sample + P.sample_base == 0u) ? 0x80000000u : 0u);
        }
        RTOW_COUNT(14, 1);
        e_next += n_gen;
        pool_n = n_gen;
        pool_head = 0;
        // the pool is written and read by different lanes of this one wave: LDS serves a wave's
        // requests in order, the compiler only has to keep them in order too
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      // ---- hand pooled paths to the idle lanes
      const uint32_t avail = pool_n - pool_head;
      const uint32_t rank = lanes_below(need);
      const uint32_t cnt = (uint32_t)__popcll(need);
      const bool take = !alive && rank < avail;
      const uint32_t entry = take ? pool_head + rank : 0u;
      const uint32_t pix_word = pool[8 * 64 + entry];
      const uint32_t pix_local = pix_word & 0x7fffffffu;
      // a lane that moves on to another pixel hands its sums over first
      flush_sums(P.accum, sums, take && accpix != pix_local && accpix != 0xffffffffu, accpix, acc_r, acc_g, acc_b, lane);
      if (take) {
        if (accpix != pix_local) { accpix = pix_local; acc_r = acc_g = acc_b = 0; }
        o.x = __uint_as_float(pool[0 * 64 + entry]); o.y = __uint_as_float(pool[1 * 64 + entry]);
        o.z = __uint_as_float(pool[2 * 64 + entry]); d.x = __uint_as_float(pool[3 * 64 + entry]);
        d.y = __uint_as_float(pool[4 * 64 + entry]); d.z = __uint_as_float(pool[5 * 64 + entry]);
        g.x0 = pool[6 * 64 + entry]; g.key1 = pool[7 * 64 + entry];
        tr = tg = tb = 1.0f;
        depth = P.max_depth;
        prev = -1;
        first_sample = (pix_word >> 31) != 0u;
        alive = true;
      }
      pool_head += min(cnt, avail);
    }
    if (__ballot(alive) == 0) break;
    RTOW_COUNT(4, __popcll(__ballot(alive)));
    const unsigned long long t_seg = RTOW_CLOCK();

    // ---- one segment: world.hit + scatter (raytracer.nim:11-16)
    f3 u = {0, 0, 1};
    float tmin = 0.0f, best = INFINITY;
    int id = -1;
    if (alive) {
      ++nseg;
      segment_begin(d, u, tmin);
      for (int k = 0; k < P.n_big; ++k) {
        const BigSphere bs = P.big[k];
        float t;
        const bool ok = big_candidate(o, u, bs, tmin, bs.index == prev, t);
        offer(ok, t, bs.index, best, id);
      }
    }
    const unsigned long long t_scan0 = RTOW_CLOCK();
    // all 64 lanes take part: lanes without a path offer no candidates but help with the exact tests
    if (X_LDS && exhausted && (uint32_t)__popcll(__ballot(alive)) <= kSparseLanes)
      scan_sparse(exact_lds, nk_lds, cand, P.n_geom_pad, P.cmax, alive, o, u, tmin, prev, best, id RTOW_PROF_ARG);
    else if (X_LDS) scan_regular<K_LDS>(geom, exact_lds, nk_lds, cand, P.n_geom_pad / 4, P.cmax, alive, o, u, tmin, prev, best, id RTOW_PROF_ARG);
    else scan_regular<K_LDS>(geom, (const f4v*)P.geom_exact, nk_lds, cand, P.n_geom_pad / 4, P.cmax, alive, o, u, tmin, prev, best, id RTOW_PROF_ARG);
    const unsigned long long t_scan1 = RTOW_CLOCK();
#ifdef RTOW_PROFILE_COUNTERS
    uint32_t shade_draws = 0;
#endif
    if (alive) {
      if (P.hit_id && first_sample && depth == P.max_depth) P.hit_id[accpix] = id;

      float rr_ = 0.0f, rg_ = 0.0f, rb_ = 0.0f;
      bool ended;
      if (id < 0) {
        shade_sky(u, tr, tg, tb, rr_, rg_, rb_);
        ended = true;
      } else {
        const SphereShade sh = P.shade[id];
        f3 dn;
#ifdef RTOW_PROFILE_COUNTERS
        const uint32_t x0_before = g.x0;
#endif
        const bool scattered = shade_hit(sh, o, u, best, g, dn, tr, tg, tb);
#ifdef RTOW_PROFILE_COUNTERS
        shade_draws = (g.x0 - x0_before) * 0x144cbc89u;
#endif
        d = dn;
        prev = id;
        --depth;
        ended = !scattered || depth <= 0;  // absorbed, or depth exhausted: black (raytracer.nim:8-9,16)
      }
      if (ended) {
        acc_r += to_fix(rr_); acc_g += to_fix(rg_); acc_b += to_fix(rb_);
        alive = false;
      }
    }
    RTOW_COUNT(3, 1);
#ifdef RTOW_PROFILE_COUNTERS
    {
      uint32_t m = shade_draws;
      for (int off = 32; off > 0; off >>= 1) { const uint32_t t_ = __shfl_xor(m, off, 64); m = t_ > m ? t_ : m; }
      RTOW_COUNT(11, m / 3u);   // rejection-loop trips of the slowest lane
    }
#endif
#ifdef RTOW_PROFILE_TIMELINE
    ++iters_;
    if (lane == 0 && wave_id < 16384 && exhausted) {
      g_wave_times[12 * wave_id + 6] = t_seg; g_wave_times[12 * wave_id + 7] = t_scan0;
      g_wave_times[12 * wave_id + 8] = t_scan1; g_wave_times[12 * wave_id + 9] = RTOW_CLOCK();
      g_wave_times[12 * wave_id + 11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    RTOW_COUNT(5, t_seg - t_refill);           // refill
    RTOW_COUNT(6, t_scan0 - t_seg);            // normalise + binary64 spheres
    RTOW_COUNT(7, t_scan1 - t_scan0);          // scan (filter + exact tests)
    RTOW_COUNT(8, RTOW_CLOCK() - t_scan1);     // shading
    (void)t_refill; (void)t_seg; (void)t_scan0; (void)t_scan1;
  }

#ifdef RTOW_PROFILE_TIMELINE
  if (lane == 0 && wave_id < 16384) g_wave_times[12 * wave_id + 10] = __builtin_amdgcn_s_memrealtime();
#endif
  flush_sums(P.accum, sums, accpix != 0xffffffffu, accpix, acc_r, acc_g, acc_b, lane);
  sum_cache_drain(P.accum, sums, lane);
  RTOW_PROF_FLUSH();
#ifdef RTOW_PROFILE_TIMELINE
  if (lane == 0 && wave_id < 16384) {
    g_wave_times[12 * wave_id + 2] = __builtin_amdgcn_s_memrealtime();
    g_wave_times[12 * wave_id + 5] = iters_;
  }
#endif
  // ray-segment count: wave reduction, one atomic per wave
  const unsigned long long total = wave_sum(nseg);
  if (lane == 0) atomicAdd(P.seg_counter, total);
}

__global__ void __launch_bounds__(256) rtow_resolve_kernel(const unsigned long long* __restrict__ accum,
                                                           uint8_t* __restrict__ rgb, uint32_t n_channels,
                                                           int spp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_channels) rgb[i] = (uint8_t)quantise_fix(accum[i], spp);
}

static size_t stack_lds_bytes() { return sizeof(uint32_t) * (kTraceThreads / 64) * (kCandWords + kPoolFields * 64 + kSumCacheDwords); }
static size_t k_lds_bytes(int n_geom_pad) { return stack_lds_bytes() + sizeof(float) * (size_t)(n_geom_pad + 4); }
static size_t kx_lds_bytes(int n_geom_pad) { return k_lds_bytes(n_geom_pad) + sizeof(float4) * (size_t)n_geom_pad; }

void launch_trace(const TraceParams& p, int grid_blocks, hipStream_t stream) {
  if (p.n_geom_pad <= kMaxSpheresExactInLds)
    hipLaunchKernelGGL((rtow_trace_kernel<true, true>), dim3(grid_blocks), dim3(kTraceThreads), kx_lds_bytes(p.n_geom_pad),
                       stream, p);
  else if (p.n_geom_pad <= kMaxSpheresKInLds)
    hipLaunchKernelGGL((rtow_trace_kernel<true, false>), dim3(grid_blocks), dim3(kTraceThreads), k_lds_bytes(p.n_geom_pad),
                       stream, p);
  else
    hipLaunchKernelGGL((rtow_trace_kernel<false, false>), dim3(grid_blocks), dim3(kTraceThreads), stack_lds_bytes(), stream, p);
}

void launch_resolve(const unsigned long long* accum, uint8_t* rgb, uint32_t n_channels, int spp,
                    hipStream_t stream) {
  if (n_channels == 0) return;
  hipLaunchKernelGGL(rtow_resolve_kernel, dim3((n_channels + 255) / 256), dim3(256), 0, stream, accum, rgb,
                     n_channels, spp);
}

// ------------------------------------------------------------------ PPM text (raytracer.nim:28,35)
// writeColor prints three decimal integers and a newline per pixel.  At 4K that is ~100 MB of
// text, seconds of fprintf on the host; on the device it is a byte-shuffling pass bounded by HBM
// (3 bytes read, 6-12 written per pixel): lengths per workgroup, a prefix sum, then every
// workgroup composes its lines in LDS and stores them as one contiguous run.
static __device__ __forceinline__ uint32_t dec_digits(uint32_t v) { return v >= 100u ? 3u : (v >= 10u ? 2u : 1u); }

static __device__ __forceinline__ uint32_t pixel_text_len(const uint8_t* rgb, uint32_t pix, uint32_t n_pixels,
                                                          uint32_t& r, uint32_t& g, uint32_t& b) {
  if (pix >= n_pixels) { r = g = b = 0; return 0u; }
  r = rgb[3ull * pix]; g = rgb[3ull * pix + 1]; b = rgb[3ull * pix + 2];
  return dec_digits(r) + dec_digits(g) + dec_digits(b) + 3u;   // two blanks and the newline
}

// inclusive prefix sum over the kPpmBlock threads of a workgroup; returns this thread's inclusive
// value, `total` = sum over the workgroup
static __device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(v, off, 64);
    if ((int)lane >= off) v += up;
  }
  if (lane == 63u) wave_tot[wave] = v;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kPpmBlock / 64; ++w) {
    const uint32_t t = wave_tot[w];
    before += w < wave ? t : 0u;
    total += t;
  }
  __syncthreads();
  return v + before;
}

__global__ void __launch_bounds__(kPpmBlock) ppm_lengths_kernel(const uint8_t* __restrict__ rgb, uint32_t n_pixels,
                                                               uint32_t* __restrict__ block_len) {
  __shared__ uint32_t wave_tot[kPpmBlock / 64];
  uint32_t r, g, b, total;
  const uint32_t len = pixel_text_len(rgb, blockIdx.x * kPpmBlock + threadIdx.x, n_pixels, r, g, b);
  (void)block_inclusive_scan(len, wave_tot, total);
  if (threadIdx.x == 0) block_len[blockIdx.x] = total;
}

// one workgroup of 1024 threads: thread t owns a contiguous span of the entries
__global__ void __launch_bounds__(1024) ppm_scan_kernel(const uint32_t* __restrict__ block_len, uint32_t n_blocks,
                                                         unsigned long long* __restrict__ block_off) {
  __shared__ unsigned long long span_sum[1024];
  const uint32_t per = (n_blocks + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
  unsigned long long s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += block_len[i];
  span_sum[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; ++i) { const unsigned long long t = span_sum[i]; span_sum[i] = run; run += t; }
    block_off[n_blocks] = run;
  }
  __syncthreads();
  unsigned long long run = span_sum[threadIdx.x];
  for (uint32_t i = lo; i < hi; ++i) { block_off[i] = run; run += block_len[i]; }
}

static __device__ __forceinline__ uint32_t put_decimal(char* dst, uint32_t at, uint32_t v, char after) {
  if (v >= 100u) dst[at++] = (char)('0' + v / 100u);
  if (v >= 10u) dst[at++] = (char)('0' + (v / 10u) % 10u);
  dst[at++] = (char)('0' + v % 10u);
  dst[at++] = after;
  return at;
}

__global__ void __launch_bounds__(kPpmBlock) ppm_text_kernel(const uint8_t* __restrict__ rgb, uint32_t n_pixels,
                                                            const unsigned long long* __restrict__ block_off,
                                                            char* __restrict__ text) {
  __shared__ uint32_t wave_tot[kPpmBlock / 64];
  __shared__ char lines[kPpmBlock * 12];
  uint32_t r, g, b, total;
  const uint32_t len = pixel_text_len(rgb, blockIdx.x * kPpmBlock + threadIdx.x, n_pixels, r, g, b);
  uint32_t at = block_inclusive_scan(len, wave_tot, total) - len;
  if (len) {
    at = put_decimal(lines, at, r, ' ');
    at = put_decimal(lines, at, g, ' ');
    (void)put_decimal(lines, at, b, '\n');
  }
  __syncthreads();
  char* const dst = text + block_off[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < total; i += kPpmBlock) dst[i] = lines[i];
}

void launch_ppm_lengths(const uint8_t* rgb, uint32_t n_pixels, uint32_t* block_len, hipStream_t stream) {
  if (n_pixels == 0) return;
  hipLaunchKernelGGL(ppm_lengths_kernel, dim3((n_pixels + kPpmBlock - 1) / kPpmBlock), dim3(kPpmBlock), 0, stream, rgb,
                     n_pixels, block_len);
}

void launch_ppm_scan(const uint32_t* block_len, uint32_t n_blocks, unsigned long long* block_off, hipStream_t stream) {
  hipLaunchKernelGGL(ppm_scan_kernel, dim3(1), dim3(1024), 0, stream, block_len, n_blocks, block_off);
}

void launch_ppm_text(const uint8_t* rgb, uint32_t n_pixels, const unsigned long long* block_off, char* text,
                     hipStream_t stream) {
  if (n_pixels == 0) return;
  hipLaunchKernelGGL(ppm_text_kernel, dim3((n_pixels + kPpmBlock - 1) / kPpmBlock), dim3(kPpmBlock), 0, stream, rgb,
                     n_pixels, block_off, text);
}

// ------------------------------------------------------------------ self test of rt_sqrt / rt_rcp
// every binary32 bit pattern against the compiler's correctly rounded sqrtf(x) and 1.0f / x
__global__ void __launch_bounds__(256) arithmetic_selftest_kernel(unsigned long long* mismatches) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned int bad_sqrt = 0, bad_rcp = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float a = rt_sqrt(x), b = sqrtf(x), c = rt_rcp(x), d = 1.0f / x;
    bad_sqrt += !(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b));
    bad_rcp += !(__float_as_uint(c) == __float_as_uint(d) || (c != c && d != d));
  }
  if (bad_sqrt) atomicAdd(&mismatches[0], (unsigned long long)bad_sqrt);
  if (bad_rcp) atomicAdd(&mismatches[1], (unsigned long long)bad_rcp);
}

void launch_arithmetic_selftest(unsigned long long* mismatches, hipStream_t stream) {
  hipLaunchKernelGGL(arithmetic_selftest_kernel, dim3(256 * 8), dim3(256), 0, stream, mismatches);
}

#ifdef RTOW_PROFILE_COUNTERS
extern "C" int rtow_debug_counters(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -2;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#endif

#ifdef RTOW_PROFILE_TIMELINE
extern "C" int rtow_debug_wave_times(unsigned long long* out, int n_waves) {
  if (n_waves > 16384) n_waves = 16384;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 12 * n_waves) != hipSuccess) return -2;
  return 0;
}
#endif

int trace_blocks_per_cu(int n_geom_pad) {
  int nb = 0;
  hipError_t e;
  if (n_geom_pad <= kMaxSpheresExactInLds)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<true, true>, kTraceThreads, kx_lds_bytes(n_geom_pad));
  else if (n_geom_pad <= kMaxSpheresKInLds)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<true, false>, kTraceThreads, k_lds_bytes(n_geom_pad));
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<false, false>, kTraceThreads, stack_lds_bytes());
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}

}  // namespace rtow
