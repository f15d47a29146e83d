// Exhaustive check (all 2^32 binary32 bit patterns) of short correctly-rounded sqrt / reciprocal
// sequences against the compiler's IEEE sqrtf and 1.0f / x on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off bench/ubench_lean_math.hip -o bench/ubench_lean_math
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

__device__ __forceinline__ float lean_sqrt(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = fmaf(-s_dn, s, x), r_up = fmaf(-s_up, s, x);
  s = r_dn <= 0.0f ? s_dn : s;
  s = r_up > 0.0f ? s_up : s;
  return s;
}
__device__ __forceinline__ float lean_rcp1(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  const float e = fmaf(-x, y, 1.0f);
  return fmaf(e, y, y);
}
__device__ __forceinline__ float lean_rcp2(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, y, 1.0f);
  y = fmaf(e, y, y);
  e = fmaf(-x, y, 1.0f);
  return fmaf(e, y, y);
}
__device__ __forceinline__ bool same(float a, float b) {
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}
// counts[k]: mismatches; lo[k]/hi[k]: smallest/largest |x| bit pattern (sign cleared) that mismatched, positive finite inputs only
__global__ void check(unsigned long long* counts, unsigned int* lo, unsigned int* hi) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float ref_s = sqrtf(x), ref_r = 1.0f / x;
    const float got[3] = {lean_sqrt(x), lean_rcp1(x), lean_rcp2(x)};
    const float ref[3] = {ref_s, ref_r, ref_r};
    for (int k = 0; k < 3; ++k)
      if (!same(got[k], ref[k])) {
        atomicAdd(&counts[k], 1ull);
        const uint32_t a = (uint32_t)i & 0x7fffffffu;
        if (k == 0 && (i >> 31)) continue;  // negative inputs of sqrt: NaN either way, payloads may differ
        if (a == 0u || a >= 0x7f800000u) continue;  // zero, infinities, NaNs: reported in the total only
        // lo: largest mismatching |x| below 1, hi: smallest mismatching |x| above 1
        if (a < 0x3f800000u) atomicMax(&lo[k], a); else atomicMin(&hi[k], a);
        // mismatches inside the comfortable range are what matters
        if (a >= 0x10000000u && a <= 0x6f000000u) atomicAdd(&counts[3 + k], 1ull);
      }
  }
}
int main() {
  unsigned long long* c; unsigned int *lo, *hi;
  hipMalloc(&c, 64); hipMalloc(&lo, 16); hipMalloc(&hi, 16);
  hipMemset(c, 0, 64); hipMemset(lo, 0, 16); hipMemset(hi, 0xff, 16);
  check<<<256 * 8, 256>>>(c, lo, hi);
  unsigned long long hc[8]; unsigned int hl[4], hh[4];
  hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost); hipMemcpy(hl, lo, 16, hipMemcpyDeviceToHost); hipMemcpy(hh, hi, 16, hipMemcpyDeviceToHost);
  const char* names[3] = {"lean_sqrt", "lean_rcp1 (one refinement)", "lean_rcp2 (two refinements)"};
  for (int k = 0; k < 3; ++k) {
    float flo, fhi; memcpy(&flo, &hl[k], 4); memcpy(&fhi, &hh[k], 4);
    printf("%-28s mismatches %llu  (inside [2^-95, 2^95]: %llu)  largest mismatching |x| < 1: %08x (%g), smallest > 1: %08x (%g)\n", names[k], hc[k],
           hc[3 + k], hl[k], flo, hh[k], fhi);
  }
  return 0;
}
