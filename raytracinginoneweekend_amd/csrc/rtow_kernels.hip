// gfx950 kernels of the renderer.
//
//  rtow_trace_kernel   persistent waves; every lane carries one path (pixel, sample) through an
//                      iterative bounce loop and is refilled with a new path the moment its
//                      path ends, so the sphere scan always runs with full wavefronts.  Work is
//                      handed out in chunks (a run of samples of one pixel) from a global
//                      counter.  Radiance is summed per lane in 64-bit fixed point and flushed
//                      with integer atomics when the lane moves to another pixel, which makes
//                      the image independent of scheduling, chunking and GPU count.
//  rtow_resolve_kernel mean + sqrt gamma + clamp + 8-bit quantisation (raytracer.nim:22-28).
//
// The sphere scan (hittables.nim:79-88 over hitrecord.nim:55-63) keeps the ray in VGPRs and
// streams the sphere table through SGPRs with wave-uniform scalar loads, four spheres per load,
// two spheres per packed FP32 instruction, and a rarely taken branch for discriminant >= 0.
#include <hip/hip_runtime.h>
#include "rtow_kernels.h"

namespace rtow {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
// constant address space: wave-uniform loads from it become s_load_dwordx16 into SGPRs
typedef const f16v __attribute__((address_space(4)))* geom_ptr;

static __device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Closest hit over the regular (binary32) table.
//
// The table is stored in groups of four spheres, 16 floats per group, as two packed pairs:
//   cx0 cx1  cy0 cy1  cz0 cz1  rr0 rr1 | cx2 cx3  cy2 cy3  cz2 cz3  rr2 rr3     (rr = r^2)
// One s_load_dwordx16 brings a group into SGPRs; the next group is requested before the current
// one is tested.  Each pair is tested with packed FP32 instructions (v_pk_add/mul/fma_f32: two
// spheres per instruction, the ray component broadcast to both halves), and one comparison on
// the maximum of the four discriminants decides whether any lane has a candidate.  Entries of
// spheres handled elsewhere and the padding carry r^2 = -inf, so their discriminant is never >= 0.
// Per sphere the arithmetic is exactly: oc = o - c; b = fma(oc.z,u.z, fma(oc.y,u.y, oc.x*u.x));
// disc = fma(b,b, fma(-oc.z,oc.z, fma(-oc.y,oc.y, fma(-oc.x,oc.x, r^2)))).
static __device__ __forceinline__ void pair_test(const f2v cx, const f2v cy, const f2v cz, const f2v rr,
                                                 const f2v ox, const f2v oy, const f2v oz, const f2v ux,
                                                 const f2v uy, const f2v uz, f2v& b, f2v& disc) {
  const f2v ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
  b = ocx * ux;
  b = __builtin_elementwise_fma(ocy, uy, b);
  b = __builtin_elementwise_fma(ocz, uz, b);
  f2v m = __builtin_elementwise_fma(-ocx, ocx, rr);
  m = __builtin_elementwise_fma(-ocy, ocy, m);
  m = __builtin_elementwise_fma(-ocz, ocz, m);
  disc = __builtin_elementwise_fma(b, b, m);
}

static __device__ __forceinline__ void scan_regular(geom_ptr geom, int n_groups, f3 o, f3 u, float tmin,
                                                    int prev, float& best, int& id) {
  if (n_groups <= 0) return;
  const f2v ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f2v ux = {u.x, u.x}, uy = {u.y, u.y}, uz = {u.z, u.z};
  f16v cur = geom[0];
  for (int g = 0; g < n_groups; ++g) {
    const f16v nxt = geom[g + 1 < n_groups ? g + 1 : g];
    f2v b0, d0, b1, d1;
    pair_test(cur.s01, cur.s23, cur.s45, cur.s67, ox, oy, oz, ux, uy, uz, b0, d0);
    pair_test(cur.s89, cur.sab, cur.scd, cur.sef, ox, oy, oz, ux, uy, uz, b1, d1);
    const float mx = fmaxf(fmaxf(d0.x, d0.y), fmaxf(d1.x, d1.y));
    if (mx >= 0.0f) {
      const int j = 4 * g;
      float t;
      if (d0.x >= 0.0f && regular_candidate(b0.x, d0.x, tmin, j == prev, t)) offer(t, j, best, id);
      if (d0.y >= 0.0f && regular_candidate(b0.y, d0.y, tmin, j + 1 == prev, t)) offer(t, j + 1, best, id);
      if (d1.x >= 0.0f && regular_candidate(b1.x, d1.x, tmin, j + 2 == prev, t)) offer(t, j + 2, best, id);
      if (d1.y >= 0.0f && regular_candidate(b1.y, d1.y, tmin, j + 3 == prev, t)) offer(t, j + 3, best, id);
    }
    cur = nxt;
  }
}

__global__ void __launch_bounds__(kTraceThreads) rtow_trace_kernel(const TraceParams P) {
  const uint32_t lane = threadIdx.x & 63u;
  const geom_ptr geom = (geom_ptr)(uintptr_t)P.geom;

  // wave-uniform: the chunk being handed out.  A chunk is a run of chunk_paths consecutive paths
  // in pixel-major order (all samples of a pixel, then the next pixel of the row list).
  uint32_t e_next = 0, e_end = 0;     // offsets inside the chunk
  uint32_t p_base = 0, s_base = 0;    // pixel and sample of the chunk's first path
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

  for (;;) {
    // ---- refill idle lanes
    for (;;) {
      const uint64_t need = __ballot(!alive);
      if (need == 0) break;
      if (e_next >= e_end) {
        if (exhausted) break;
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(P.chunk_counter, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= P.n_chunks) { exhausted = true; break; }
        const unsigned long long first = (unsigned long long)c * P.chunk_paths;
        const unsigned long long left = P.total_paths - first;
        p_base = (uint32_t)(first / (uint32_t)P.spp);
        s_base = (uint32_t)(first - (unsigned long long)p_base * (uint32_t)P.spp);
        e_next = 0;
        e_end = left < P.chunk_paths ? (uint32_t)left : P.chunk_paths;
      }
      const uint32_t avail = e_end - e_next;
      const uint32_t rank = lanes_below(need);
      const uint32_t cnt = (uint32_t)__popcll(need);
      if (!alive && rank < avail) {
        const uint32_t e = s_base + e_next + rank;
        const uint32_t dp = e / (uint32_t)P.spp;
        const uint32_t sample = e - dp * (uint32_t)P.spp;
        const uint32_t pix_local = p_base + dp;
        if (accpix != pix_local) {
          if (accpix != 0xffffffffu) {
            atomicAdd(&P.accum[3ull * accpix + 0], acc_r);
            atomicAdd(&P.accum[3ull * accpix + 1], acc_g);
            atomicAdd(&P.accum[3ull * accpix + 2], acc_b);
          }
          accpix = pix_local; acc_r = acc_g = acc_b = 0;
        }
        const uint32_t r = pix_local / (uint32_t)P.W;
        const int ci = (int)(pix_local - r * (uint32_t)P.W);
        const int y = P.rows ? P.rows[r] : (int)r;
        g = stream_begin(P.seed_lo, P.seed_hi, (uint32_t)y * (uint32_t)P.W + (uint32_t)ci, sample);
        camera_ray(P.cam, ci, P.H - 1 - y, P.W, P.H, g, o, d);
        tr = tg = tb = 1.0f;
        depth = P.max_depth;
        prev = -1;
        first_sample = (sample == 0);
        alive = true;
      }
      e_next += min(cnt, avail);
    }
    if (__ballot(alive) == 0) break;

    // ---- one segment: world.hit + scatter (raytracer.nim:11-16)
    if (alive) {
      ++nseg;
      f3 u; float tmin;
      segment_begin(d, u, tmin);
      float best = INFINITY; int id = -1;
      for (int k = 0; k < P.n_big; ++k) {
        const BigSphere bs = P.big[k];
        float t;
        if (big_candidate(o, u, bs, tmin, bs.index == prev, t)) offer(t, bs.index, best, id);
      }
      scan_regular(geom, P.n_geom_pad / 4, o, u, tmin, prev, best, id);

      if (P.hit_id && first_sample && depth == P.max_depth) P.hit_id[accpix] = id;

      float rr = 0.0f, rg = 0.0f, rb = 0.0f;
      bool ended;
      if (id < 0) {
        shade_sky(u, tr, tg, tb, rr, rg, rb);
        ended = true;
      } else {
        const SphereShade sh = P.shade[id];
        f3 dn;
        const bool scattered = shade_hit(sh, o, u, best, g, dn, tr, tg, tb);
        d = dn;
        prev = id;
        --depth;
        ended = !scattered || depth <= 0;  // absorbed, or depth exhausted: black (raytracer.nim:8-9,16)
      }
      if (ended) {
        acc_r += to_fix(rr); acc_g += to_fix(rg); acc_b += to_fix(rb);
        alive = false;
      }
    }
  }

  if (accpix != 0xffffffffu) {
    atomicAdd(&P.accum[3ull * accpix + 0], acc_r);
    atomicAdd(&P.accum[3ull * accpix + 1], acc_g);
    atomicAdd(&P.accum[3ull * accpix + 2], acc_b);
  }
  // ray-segment count: wave reduction, one atomic per wave
  unsigned long long total = nseg;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
  if (lane == 0) atomicAdd(P.seg_counter, total);
}

__global__ void __launch_bounds__(256) rtow_resolve_kernel(const unsigned long long* __restrict__ accum,
                                                           uint8_t* __restrict__ rgb, uint32_t n_channels,
                                                           int spp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_channels) rgb[i] = (uint8_t)quantise_fix(accum[i], spp);
}

void launch_trace(const TraceParams& p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(rtow_trace_kernel, dim3(grid_blocks), dim3(kTraceThreads), 0, stream, p);
}

void launch_resolve(const unsigned long long* accum, uint8_t* rgb, uint32_t n_channels, int spp,
                    hipStream_t stream) {
  if (n_channels == 0) return;
  hipLaunchKernelGGL(rtow_resolve_kernel, dim3((n_channels + 255) / 256), dim3(256), 0, stream, accum, rgb,
                     n_channels, spp);
}

int trace_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel, kTraceThreads, 0) != hipSuccess || nb < 1)
    nb = 1;
  return nb;
}

}  // namespace rtow
