// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lbmk.h"

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct __attribute__((aligned(16))) u16x8 { bf16_t v[8]; };
struct __attribute__((aligned(8))) u16x4 { bf16_t v[4]; };

__device__ __forceinline__ float bf2f(bf16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

// x * sigmoid(x) on the hardware exp2 / rcp (1 ulp each): an IEEE divide costs ~10 VALU ops per element,
// which turns the bandwidth-bound GroupNorm+SiLU pass into a VALU-bound one
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below one bf16 ulp of the GELU output) on the hardware
// exp2 / rcp: about a third of the vector instructions of the library erff, which a GEGLU epilogue runs 32 times
// per lane and tile
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(1.0f - poly * t * e, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// async 16-byte global -> LDS copy; lds_wave_base must be wave-uniform, lane i lands at base + 16*i
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 16-byte LDS-DMA load through a buffer descriptor over [base, base + bytes): lane l's data lands at
// lds_wave_base + 16 * l; a byte offset (voff + soff) at or past `bytes` yields zeros.  base / bytes /
// soff must be wave-uniform.
__device__ __forceinline__ void buffer_lds16(const void* base, int bytes, unsigned voff, unsigned soff,
                                             void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000),
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// host side
void lbmk_set_error(const char* fmt, ...);
#define LBMK_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      lbmk_set_error(__VA_ARGS__);         \
      return LBMK_ERR_ARG;                 \
    }                                      \
  } while (0)
int lbmk_check_launch(const char* what);
// compute units of the current device (cached per device; 256 when the query fails)
int lbmk_num_cus(void);
static inline bool lbmk_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }
