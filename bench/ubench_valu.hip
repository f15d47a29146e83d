// Micro-benchmarks that decide the shape of the ray-sphere hot loop on gfx950.
//
//  (1) raw issue rate of v_fma_f32 vs v_pk_fma_f32 at 1..8 waves per SIMD
//  (2) the closest-hit scan itself, sphere operands fed three ways:
//        SCALAR  - uniform s_load from global memory (SGPR operands)
//        LDS     - uniform ds_read_b128 broadcast (VGPR operands)
//        PACKED  - two spheres per v_pk_* instruction, SGPR pairs
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off bench/ubench_valu.hip -o bench/ubench_valu
// Run:   bench/ubench_valu            (prints one line per variant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------- raw rates
// MODE 0: v_fma_f32 (VOP3, 8-byte encoding)   1: v_pk_fma_f32   2: v_fmac_f32_e32 (VOP2, 4-byte)
// MODE 3: the 11-instruction test body (3 sub, mul, 2 fmac, 3 fma, fmac, cmp) on SGPR sphere operands
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, unsigned long long* clk) {
  float a0 = threadIdx.x * 1e-9f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 0.999f + a0, c = 1e-6f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 q0 = p0 + 8.f, q1 = p1 + 8.f, q2 = p2 + 8.f, q3 = p3 + 8.f;
  f2 mm = {m, m}, cc = {c, c};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n"
            "v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
            "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "v"(mm), "v"(cc));
      }
    } else if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
            "v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
            "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(m), "v"(c));
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n"
            "v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
            "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n"
            "v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(m), "v"(c));
      }
    } else if (MODE == 4) {
      // v_pk_fma_f32 with an SGPR-pair source (the filter's operand mix)
      const f2 sp = {__builtin_amdgcn_readfirstlane(i) * 1e-9f + 0.999f, 0.998f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %8, %0, %9\n v_pk_fma_f32 %1, %8, %1, %9\n"
            "v_pk_fma_f32 %2, %8, %2, %9\n v_pk_fma_f32 %3, %8, %3, %9\n"
            "v_pk_fma_f32 %4, %8, %4, %9\n v_pk_fma_f32 %5, %8, %5, %9\n"
            "v_pk_fma_f32 %6, %8, %6, %9\n v_pk_fma_f32 %7, %8, %7, %9\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "s"(sp), "v"(cc));
      }
    } else if (MODE == 7) {
      // SGPR pair as src1 instead of src0
      const f2 sp = {__builtin_amdgcn_readfirstlane(i) * 1e-9f + 0.999f, 0.998f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n"
            "v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
            "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
            "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "s"(sp), "v"(cc));
      }
    } else if (MODE == 8) {
      // SGPR pair as src2 (addend)
      const f2 sp = {__builtin_amdgcn_readfirstlane(i) * 1e-9f + 1e-6f, 1e-6f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %0, %9, %8\n v_pk_fma_f32 %1, %1, %9, %8\n"
            "v_pk_fma_f32 %2, %2, %9, %8\n v_pk_fma_f32 %3, %3, %9, %8\n"
            "v_pk_fma_f32 %4, %4, %9, %8\n v_pk_fma_f32 %5, %5, %9, %8\n"
            "v_pk_fma_f32 %6, %6, %9, %8\n v_pk_fma_f32 %7, %7, %9, %8\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "s"(sp), "v"(mm));
      }
    } else if (MODE == 9) {
      // plain v_fma_f32 with an SGPR source
      const float sp = __builtin_amdgcn_readfirstlane(i) * 1e-9f + 0.999f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n"
            "v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
            "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n"
            "v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "s"(sp), "v"(c));
      }
    } else if (MODE == 10) {
      // SGPR source read as ONE dword for both halves (op_sel / op_sel_hi both 0 on src0)
      const f2 sp = {__builtin_amdgcn_readfirstlane(i) * 1e-9f + 0.999f, 0.998f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %8, %0, %9 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %8, %1, %9 op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 %2, %8, %2, %9 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %8, %3, %9 op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 %4, %8, %4, %9 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5, %8, %5, %9 op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 %6, %8, %6, %9 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7, %8, %7, %9 op_sel_hi:[0,1,1]\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "s"(sp), "v"(cc));
      }
    } else if (MODE == 11) {
      // SGPR pair source, the two VGPR sources the SAME register pair (fewer VGPR reads per instruction)
      const f2 sp = {__builtin_amdgcn_readfirstlane(i) * 1e-9f + 0.999f, 0.998f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_fma_f32 %0, %8, %0, %0\n v_pk_fma_f32 %1, %8, %1, %1\n"
            "v_pk_fma_f32 %2, %8, %2, %2\n v_pk_fma_f32 %3, %8, %3, %3\n"
            "v_pk_fma_f32 %4, %8, %4, %4\n v_pk_fma_f32 %5, %8, %5, %5\n"
            "v_pk_fma_f32 %6, %8, %6, %6\n v_pk_fma_f32 %7, %8, %7, %7\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "s"(sp), "v"(cc));
      }
    } else if (MODE == 5) {
      // v_pk_mul_f32 / v_pk_add_f32 mix
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n"
            "v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
            "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n"
            "v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
            : "v"(mm), "v"(cc));
      }
    } else if (MODE == 6) {
      // integer multiply rate
      unsigned x0 = __float_as_uint(a0), x1 = __float_as_uint(a1), x2 = __float_as_uint(a2), x3 = __float_as_uint(a3);
      unsigned x4 = __float_as_uint(a4), x5 = __float_as_uint(a5), x6 = __float_as_uint(a6), x7 = __float_as_uint(a7);
      unsigned mk = 0x7feb352du;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n"
            "v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
            "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n"
            "v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
            : "v"(mk));
      }
      a0 = __uint_as_float(x0); a1 = __uint_as_float(x1); a2 = __uint_as_float(x2); a3 = __uint_as_float(x3);
      a4 = __uint_as_float(x4); a5 = __uint_as_float(x5); a6 = __uint_as_float(x6); a7 = __uint_as_float(x7);
    } else {
      // 32 VALU instructions ~ 3 test bodies, no branch: ray in a1..a6 (o,d), sphere in s[..] via "s" operands
      float sx = __builtin_amdgcn_readfirstlane(i) * 1e-9f, sw = 0.04f;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        asm volatile(
            "v_subrev_f32_e32 %0, %8, %2\n v_subrev_f32_e32 %1, %8, %3\n"
            "v_mul_f32_e32 %6, %5, %0\n v_fma_f32 %0, -%0, %0, %9\n"
            "v_subrev_f32_e32 %7, %8, %4\n v_fmac_f32_e32 %6, %1, %5\n"
            "v_fma_f32 %0, -%1, %1, %0\n v_fmac_f32_e32 %6, %7, %5\n"
            "v_fma_f32 %0, -%7, %7, %0\n v_fmac_f32_e32 %0, %6, %6\n"
            "v_cmp_le_f32_e32 vcc, 0, %0\n"
            : "=&v"(a0), "=&v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "=&v"(a6), "=&v"(a7)
            : "s"(sx), "s"(sw) : "vcc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
            p3.x + p3.y + q0.x + q0.y + q1.x + q1.y + q2.x + q2.y + q3.x + q3.y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ------------------------------------------------------- closest-hit scans
struct Hit { float t; int id; };

__device__ __forceinline__ void hit_update(float b, float disc, float tmin, int j, float& best, int& id) {
  float s = sqrtf(disc);
  float t0 = -b - s;
  if (t0 < tmin || best < t0) {
    t0 = -b + s;
    if (t0 < tmin || best < t0) return;
  }
  best = t0; id = j;
}

enum { V_SCALAR = 0, V_LDS = 1, V_PACKED = 2, V_SCALAR_MAX3 = 3, V_SCALAR_PF = 4 };
typedef float f16v __attribute__((ext_vector_type(16)));

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4)))* cf4ptr;

template <int VAR, int MAXN>
__global__ void __launch_bounds__(256) scan_kernel(const float4* __restrict__ spheres, int n,
                                                   const float* __restrict__ rays, int nrays,
                                                   Hit* __restrict__ out, int reps) {
  __shared__ float4 lds[VAR == V_LDS ? MAXN : 1];
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (VAR == V_LDS) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = spheres[i];
    __syncthreads();
  }
  int r = tid % nrays;
  float ox = rays[0 * nrays + r], oy = rays[1 * nrays + r], oz = rays[2 * nrays + r];
  float dx = rays[3 * nrays + r], dy = rays[4 * nrays + r], dz = rays[5 * nrays + r];
  float acc_t = 0.f; int acc_id = 0;
  cf4ptr sp = (cf4ptr)(uintptr_t)spheres;
  for (int rep = 0; rep < reps; ++rep) {
    float best = INFINITY; int id = -1;
    const float tmin = 0.001f;
    if (VAR == V_SCALAR) {
#pragma unroll 4
      for (int j = 0; j < n; ++j) {
        f4v s = sp[j];
        float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
        float b = ocx * dx; b = fmaf(ocy, dy, b); b = fmaf(ocz, dz, b);
        float m = fmaf(-ocx, ocx, s.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
        float disc = fmaf(b, b, m);
        if (disc >= 0.f) hit_update(b, disc, tmin, j, best, id);
      }
    } else if (VAR == V_SCALAR_MAX3) {
      for (int j0 = 0; j0 < n; j0 += 4) {
        float dsc[4], bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f4v s = sp[j0 + u];
          float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
          float b = ocx * dx; b = fmaf(ocy, dy, b); b = fmaf(ocz, dz, b);
          float m = fmaf(-ocx, ocx, s.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
          dsc[u] = fmaf(b, b, m); bb[u] = b;
        }
        float mx = fmaxf(fmaxf(dsc[0], dsc[1]), fmaxf(dsc[2], dsc[3]));
        if (mx >= 0.f) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (dsc[u] >= 0.f) hit_update(bb[u], dsc[u], tmin, j0 + u, best, id);
        }
      }
    } else if (VAR == V_SCALAR_PF) {
      // software-pipelined: one s_load_dwordx16 (4 spheres) in flight while the previous 4 are tested
      typedef const f16v __attribute__((address_space(4)))* cf16ptr;
      cf16ptr sp16 = (cf16ptr)(uintptr_t)spheres;
      f16v cur = sp16[0];
      for (int g = 0; g < n / 4; ++g) {
        f16v nxt = sp16[g + 1 < n / 4 ? g + 1 : g];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float sx = cur[4 * u], sy = cur[4 * u + 1], sz = cur[4 * u + 2], sw = cur[4 * u + 3];
          float ocx = ox - sx, ocy = oy - sy, ocz = oz - sz;
          float b = ocx * dx; b = fmaf(ocy, dy, b); b = fmaf(ocz, dz, b);
          float m = fmaf(-ocx, ocx, sw); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
          float disc = fmaf(b, b, m);
          if (disc >= 0.f) hit_update(b, disc, tmin, 4 * g + u, best, id);
        }
        cur = nxt;
      }
    } else if (VAR == V_LDS) {
#pragma unroll 4
      for (int j = 0; j < n; ++j) {
        float4 s = lds[j];
        float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
        float b = ocx * dx; b = fmaf(ocy, dy, b); b = fmaf(ocz, dz, b);
        float m = fmaf(-ocx, ocx, s.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
        float disc = fmaf(b, b, m);
        if (disc >= 0.f) hit_update(b, disc, tmin, j, best, id);
      }
    } else if (VAR == V_PACKED) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      // sphere pairs stored as (cx0,cx1),(cy0,cy1) | (cz0,cz1),(r2_0,r2_1): two float4 per pair
      f2 oxx = {ox, ox}, oyy = {oy, oy}, ozz = {oz, oz};
      f2 dxx = {dx, dx}, dyy = {dy, dy}, dzz = {dz, dz};
#pragma unroll 2
      for (int j = 0; j < n; j += 2) {
        f4v s0 = sp[j], s1 = sp[j + 1];
        f2 cx = {s0.x, s0.y}, cy = {s0.z, s0.w}, cz = {s1.x, s1.y}, r2 = {s1.z, s1.w};
        f2 ocx = oxx - cx, ocy = oyy - cy, ocz = ozz - cz;
        f2 b = ocx * dxx;
        b = __builtin_elementwise_fma(ocy, dyy, b);
        b = __builtin_elementwise_fma(ocz, dzz, b);
        f2 m = __builtin_elementwise_fma(-ocx, ocx, r2);
        m = __builtin_elementwise_fma(-ocy, ocy, m);
        m = __builtin_elementwise_fma(-ocz, ocz, m);
        f2 disc = __builtin_elementwise_fma(b, b, m);
        if (fmaxf(disc.x, disc.y) >= 0.f) {
          if (disc.x >= 0.f) hit_update(b.x, disc.x, tmin, j, best, id);
          if (disc.y >= 0.f) hit_update(b.y, disc.y, tmin, j + 1, best, id);
        }
      }
    }
    acc_t += (id >= 0 ? best : 0.f); acc_id += id;
    ox += 1e-4f; oz -= 1e-4f;  // keep the scan loop-variant
  }
  out[tid].t = acc_t; out[tid].id = acc_id;
}

static double run_timed(void (*launch)(void*), void* ctx, int nrep) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(ctx); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < nrep; ++i) launch(ctx);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / nrep * 1e-3;
}

struct RateCtx { float* out; int iters; int blocks; int threads; int mode; unsigned long long* clk; };
static void launch_rate(void* p) {
  RateCtx* c = (RateCtx*)p;
  dim3 g(c->blocks), b(c->threads);
  switch (c->mode) {
    case 0: hipLaunchKernelGGL(rate_kernel<0>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 1: hipLaunchKernelGGL(rate_kernel<1>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 2: hipLaunchKernelGGL(rate_kernel<2>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 3: hipLaunchKernelGGL(rate_kernel<3>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 4: hipLaunchKernelGGL(rate_kernel<4>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 5: hipLaunchKernelGGL(rate_kernel<5>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 6: hipLaunchKernelGGL(rate_kernel<6>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 7: hipLaunchKernelGGL(rate_kernel<7>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 8: hipLaunchKernelGGL(rate_kernel<8>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 9: hipLaunchKernelGGL(rate_kernel<9>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 10: hipLaunchKernelGGL(rate_kernel<10>, g, b, 0, 0, c->out, c->iters, c->clk); break;
    case 11: hipLaunchKernelGGL(rate_kernel<11>, g, b, 0, 0, c->out, c->iters, c->clk); break;
  }
}

struct ScanCtx { const float4* sph; int n; const float* rays; int nrays; Hit* out; int reps; int blocks; int var; };
static void launch_scan(void* p) {
  ScanCtx* c = (ScanCtx*)p;
  dim3 g(c->blocks), b(256);
  switch (c->var) {
    case V_SCALAR: hipLaunchKernelGGL((scan_kernel<V_SCALAR, 512>), g, b, 0, 0, c->sph, c->n, c->rays, c->nrays, c->out, c->reps); break;
    case V_LDS: hipLaunchKernelGGL((scan_kernel<V_LDS, 512>), g, b, 0, 0, c->sph, c->n, c->rays, c->nrays, c->out, c->reps); break;
    case V_PACKED: hipLaunchKernelGGL((scan_kernel<V_PACKED, 512>), g, b, 0, 0, c->sph, c->n, c->rays, c->nrays, c->out, c->reps); break;
    case V_SCALAR_MAX3: hipLaunchKernelGGL((scan_kernel<V_SCALAR_MAX3, 512>), g, b, 0, 0, c->sph, c->n, c->rays, c->nrays, c->out, c->reps); break;
    case V_SCALAR_PF: hipLaunchKernelGGL((scan_kernel<V_SCALAR_PF, 512>), g, b, 0, 0, c->sph, c->n, c->rays, c->nrays, c->out, c->reps); break;
  }
}

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float urand(uint32_t& s) { return (lcg(s) >> 8) * (1.0f / 16777216.0f); }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs %d  clock %d kHz  arch %s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
  const int CU = prop.multiProcessorCount;

  // (1) raw rates
  float* d_out; CK(hipMalloc(&d_out, sizeof(float) * 256 * CU * 8));
  unsigned long long* d_clk; CK(hipMalloc(&d_clk, 16));
  const char* rn[] = {"v_fma_f32(vop3)", "v_pk_fma_f32", "v_fmac_f32(vop2)", "test-body(11)", "v_pk_fma(sgpr src)", "v_pk_mul/add", "v_mul_lo_u32", "v_pk_fma(sgpr src1)", "v_pk_fma(sgpr src2)", "v_fma_f32(sgpr src)", "v_pk_fma(sgpr dword)", "v_pk_fma(sgpr,2 vgpr)"};
  for (int mode = 0; mode < 12; ++mode)
    for (int wps = 1; wps <= 8; wps *= 2) {
      RateCtx c{d_out, 40000, CU * wps, 256, mode, d_clk};
      double s = run_timed(launch_rate, &c, 3);
      unsigned long long hclk[2]; CK(hipMemcpy(hclk, d_clk, 16, hipMemcpyDeviceToHost));
      double per = (mode == 3 ? 33 : 32);
      double instr = (double)c.blocks * 256 * c.iters * per;  // lane-instructions
      double flops = instr * 2 * ((mode == 1 || mode == 4 || mode == 7 || mode == 8 || mode == 10 || mode == 11) ? 2 : 1);
      printf("rate %-17s waves/SIMD %d : %.3f ms  %.1f Glane-instr/s  %.1f TFLOP/s  | %.2f cyc/instr/wave  clk %.0f MHz\n",
             rn[mode], wps, s * 1e3, instr / s * 1e-9, flops / s * 1e-12,
             (double)hclk[0] / (c.iters * per), (double)hclk[0] / (double)hclk[1] * 100.0);
    }

  // (2) scans: book-like scene, 484 spheres
  const int N = 484;
  std::vector<float4> sph; uint32_t seed = 12345;
  sph.push_back(make_float4(0.f, -1000.f, 0.f, 1000.f * 1000.f));
  for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
    if ((int)sph.size() >= N - 3) break;
    sph.push_back(make_float4(a + 0.9f * urand(seed), 0.2f, b + 0.9f * urand(seed), 0.04f));
  }
  while ((int)sph.size() < N - 3) sph.push_back(make_float4(30.f, 0.2f, 30.f, 0.04f));
  sph.push_back(make_float4(0, 1, 0, 1)); sph.push_back(make_float4(-4, 1, 0, 1)); sph.push_back(make_float4(4, 1, 0, 1));
  // packed layout: pairs
  std::vector<float4> sphp(N);
  for (int j = 0; j < N; j += 2) {
    sphp[j] = make_float4(sph[j].x, sph[j + 1].x, sph[j].y, sph[j + 1].y);
    sphp[j + 1] = make_float4(sph[j].z, sph[j + 1].z, sph[j].w, sph[j + 1].w);
  }
  const int NR = 1 << 16;
  std::vector<float> rays(6 * NR);
  for (int r = 0; r < NR; ++r) {
    // secondary-like rays: start on the ground plane, random upward direction
    float x = 22.f * urand(seed) - 11.f, z = 22.f * urand(seed) - 11.f;
    float dx, dy, dz, l2;
    do { dx = 2 * urand(seed) - 1; dy = urand(seed); dz = 2 * urand(seed) - 1; l2 = dx * dx + dy * dy + dz * dz; } while (l2 >= 1.f || l2 < 1e-4f);
    float il = 1.f / sqrtf(l2);
    rays[0 * NR + r] = x; rays[1 * NR + r] = 0.0005f; rays[2 * NR + r] = z;
    rays[3 * NR + r] = dx * il; rays[4 * NR + r] = dy * il; rays[5 * NR + r] = dz * il;
  }
  float4 *d_sph, *d_sphp; float* d_rays; Hit* d_hit;
  CK(hipMalloc(&d_sph, sizeof(float4) * N)); CK(hipMalloc(&d_sphp, sizeof(float4) * N));
  CK(hipMalloc(&d_rays, sizeof(float) * 6 * NR));
  CK(hipMalloc(&d_hit, sizeof(Hit) * 256 * CU * 8));
  CK(hipMemcpy(d_sph, sph.data(), sizeof(float4) * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sphp, sphp.data(), sizeof(float4) * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rays, rays.data(), sizeof(float) * 6 * NR, hipMemcpyHostToDevice));
  const char* names[] = {"scalar(s_load)", "lds-broadcast", "packed(pk)", "scalar+max4", "scalar-pipelined"};
  std::vector<Hit> h0, h1;
  for (int var = 0; var < (getenv("UBENCH_SCANS") ? 5 : 0); ++var)
    for (int wps = 1; wps <= 8; wps *= 2) {
      ScanCtx c{var == V_PACKED ? d_sphp : d_sph, N, d_rays, NR, d_hit, 512, CU * wps, var};
      double s = run_timed(launch_scan, &c, 3);
      double tests = (double)c.blocks * 256 * c.reps * N;
      printf("scan %-15s waves/SIMD %d : %.3f ms  %.2f Gtests/s  %.1f TFLOP/s(18/test)  %.1f%% of 157.3\n",
             names[var], wps, s * 1e3, tests / s * 1e-9, tests * 18 / s * 1e-12, tests * 18 / s / 157.3e12 * 100);
      if (wps == 1) {
        std::vector<Hit>& h = (var == 0 ? h0 : h1);
        h.resize(256 * CU);
        CK(hipMemcpy(h.data(), d_hit, sizeof(Hit) * 256 * CU, hipMemcpyDeviceToHost));
        if (var > 0) {
          int bad = 0;
          for (size_t i = 0; i < h.size(); ++i) if (h[i].id != h0[i].id || h[i].t != h0[i].t) ++bad;
          printf("   check vs scalar: %d mismatches of %zu\n", bad, h.size());
        }
      }
    }
  return 0;
}
