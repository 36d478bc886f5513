// Flash-style attention backward for head dims 40 / 64 / 80 / 160 on v_mfma_f32_32x32x16_bf16 (gfx950).
//
// With P = softmax(scale * Q K^T) (recomputed from the forward's per-row log-sum-exp, never stored),
// delta_i = sum_d dO[i][d] * O[i][d]  and  dS = P * (dP - delta):
//     dV = P^T dO      dP = dO V^T      dQ = scale * dS K      dK = scale * dS^T Q
// Two kernels, neither with atomics (results are bit-reproducible):
//   * dq kernel:  a workgroup owns 128 query rows (32 per wave, the row on the lane) and walks the keys;
//                 it also writes delta for the second kernel.
//   * dkv kernel: a workgroup owns 128 keys (32 per wave, the key on the lane) and walks the queries.
// As in the forward kernel, every score tile is produced transposed to the operand it feeds next, so the
// exponentiated / differentiated tile is packed to bf16 in registers and used directly as the B operand
// of the following MFMA.  Operands that are consumed with the token index along the MFMA k dimension
// (K^T for dQ, Q^T and dO^T for dK / dV) are read from channel-major copies the host passes in.
#include "common.h"

__device__ __attribute__((aligned(16))) uint32_t attn_bwd_zero_page[8];   // source of the zero padding chunks

namespace {

constexpr int RB = 128;   // rows (queries or keys) owned by a workgroup
constexpr int TB = 64;    // tokens per tile of the walked operand

// Row-major image of 64 tokens x DH channels: rows of KS*32 bytes (DH rounded up to 16, the pad chunk
// zero), read as the A operand with the channel along k.
template <int DH>
__device__ __forceinline__ void stage_rows(const bf16_t* src, int ld, char* dst, int t, int wave) {
  constexpr int KS = (DH + 15) / 16, CH = TB * KS * 2, NI = (CH + 255) / 256;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = i * 256 + t;
    if (idx < CH) {   // whole waves are in or out (CH % 64 == 0)
      const int row = idx / (KS * 2), c = idx - row * (KS * 2);
      const bf16_t* g = c * 8 < DH ? src + (int64_t)row * ld + c * 8 : (const bf16_t*)attn_bwd_zero_page;
      glds16(g, dst + (i * 256 + wave * 64) * 16);
    }
  }
}

// Channel-major image: ceil32(DH) rows (channels) x 64 tokens, 128-byte rows with the 16-byte slots
// XOR-swizzled; read as the A operand with the token along k.  Rows past DH come from the zero page.
template <int DH>
__device__ __forceinline__ void stage_cols(const bf16_t* src, int ld, char* dst, int wave, int rsub, int cs) {
  constexpr int DT = (DH + 31) / 32;
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    const int row = rsub + 32 * i;
    const bf16_t* g = row < DH ? src + (int64_t)row * ld + cs * 8 : (const bf16_t*)attn_bwd_zero_page;
    glds16(g, dst + (wave * 8 + 32 * i) * 128);
  }
}

__device__ __forceinline__ bf16x8 frag_t(const char* img, int off_lo, int off_hi) {
  const uint2 lo = *(const uint2*)(img + off_lo);
  const uint2 hi = *(const uint2*)(img + off_hi);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// lane (r,h) <- X[row][16ks + 8h .. +8], zero past DH
template <int DH>
__device__ __forceinline__ void load_row_frags(const bf16_t* rowp, int h, bf16x8 (&f)[(DH + 15) / 16]) {
#pragma unroll
  for (int ks = 0; ks < (DH + 15) / 16; ++ks) {
    if (16 * ks + 8 * h + 8 <= DH) f[ks] = *(const bf16x8*)(rowp + 16 * ks + 8 * h);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[ks][e] = (__bf16)0.f;
    }
  }
}

template <int DH>
__device__ __forceinline__ void store_rows(bf16_t* op, const f32x16 (&acc)[(DH + 31) / 32], float mul, int h) {
#pragma unroll
  for (int dt = 0; dt < (DH + 31) / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 32 * dt + 8 * g + 4 * h;
      if (d0 < DH) {
        u16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(acc[dt][4 * g + e] * mul);
        *(u16x4*)(op + d0) = ov;
      }
    }
}

template <int DH>
__global__ __launch_bounds__(256, DH > 80 ? 1 : 2) void flash_bwd_dq_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ Kt, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
    const float* __restrict__ LSE, bf16_t* __restrict__ dQ, float* __restrict__ Delta, int Nq, int Nk,
    int ld, int ldo, int ldkt, int64_t sq, int64_t sk, int64_t so, int64_t skt, float scale, float scale_log2e) {
  constexpr int KS = (DH + 15) / 16, DT = (DH + 31) / 32, KROW = KS * 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * TB * KROW + DT * 32 * 128];
  char* sK = smem;                  // [key][DH]
  char* sV = smem + TB * KROW;      // [key][DH]
  char* sKt = smem + 2 * TB * KROW; // [d][64 keys]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int q = blockIdx.x * RB + wave * 32 + r;
  const bool live = q < Nq;             // whole waves: Nq is a multiple of 64 (a dead wave only helps staging)
  const int qc = live ? q : Nq - 1;

  Q += b * sq + head * DH;
  K += b * sk + head * DH;
  V += b * sk + head * DH;
  Kt += b * skt + (int64_t)head * DH * ldkt;
  dO += b * so + head * DH;
  O += b * so + head * DH;
  dQ += b * sq + head * DH;
  const int64_t stat = ((int64_t)b * H + head) * Nq + qc;

  bf16x8 qf[KS], dof[KS];
  float delta = 0.f;
  {
    bf16x8 of[KS];
    load_row_frags<DH>(Q + (int64_t)qc * ld, h, qf);
    load_row_frags<DH>(dO + (int64_t)qc * ldo, h, dof);
    load_row_frags<DH>(O + (int64_t)qc * ldo, h, of);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += (float)dof[ks][e] * (float)of[ks][e];
    delta += __shfl_xor(delta, 32, 64);
    if (live && h == 0) Delta[stat] = delta;
  }
  const float lse = LSE[stat];

  f32x16 acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;
  const int roff = r * KROW + h * 16;
  int toff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) toff[c] = r * 128 + 8 * h + ((c ^ rswz) << 4);

  for (int kv0 = 0; kv0 < Nk; kv0 += TB) {
    stage_rows<DH>(K + (int64_t)kv0 * ld, ld, sK, t, wave);
    stage_rows<DH>(V + (int64_t)kv0 * ld, ld, sV, t, wave);
    stage_cols<DH>(Kt + kv0, ldkt, sKt, wave, rsub, cs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 s, dp;
      // S^T = K Q^T and dP^T = V dO^T for 32 keys: keys on the accumulator rows
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(sK + kt * 32 * KROW + roff + ks * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero : s, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 vf = *(const bf16x8*)(sV + kt * 32 * KROW + roff + ks * 32);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], ks == 0 ? zero : dp, 0, 0, 0);
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - lse);
        dsf[e >> 3][e & 7] = (__bf16)(p * (dp[e] - delta));
      }
      // dQ^T += K^T dS^T
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int c1 = 4 * kt + 2 * ss;
          const bf16x8 kf = frag_t(sKt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, dsf[ss], acc[dt], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (live) store_rows<DH>(dQ + (int64_t)q * ld, acc, scale, h);
}

// WANT: 3 = dK and dV in one walk; 1 = dV only, 2 = dK only.  Head dim 80 walks twice: one walk would need
// the whole register file (one wave per SIMD, nothing to overlap the staging with) and measured slower.
template <int DH, int WANT>
__global__ __launch_bounds__(256, DH > 80 ? 1 : 2) void flash_bwd_dkv_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Qt, const bf16_t* __restrict__ K,
    const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOt,
    const float* __restrict__ LSE, const float* __restrict__ Delta, bf16_t* __restrict__ dK,
    bf16_t* __restrict__ dV, int Nq, int Nk, int ld, int ldo, int ldqt, int64_t sq, int64_t sqt, int64_t sk,
    int64_t so, int64_t sdot, float scale, float scale_log2e) {
  constexpr int KS = (DH + 15) / 16, DT = (DH + 31) / 32, KROW = KS * 32;
  constexpr bool DO_DV = (WANT & 1) != 0, DO_DK = (WANT & 2) != 0;
  __shared__ __attribute__((aligned(16))) char smem[2 * TB * KROW + 2 * DT * 32 * 128 + 2 * TB * 4];
  char* sQ = smem;                                    // [query][DH]
  char* sG = smem + TB * KROW;                        // dO, [query][DH]
  char* sQt = smem + 2 * TB * KROW;                   // [d][64 queries]
  char* sGt = sQt + DT * 32 * 128;                    // dO^T, [d][64 queries]
  float* sL = (float*)(sGt + DT * 32 * 128);          // log-sum-exp of the 64 queries
  float* sD = sL + TB;                                // delta of the 64 queries

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int key = blockIdx.x * RB + wave * 32 + r;
  const bool live = key < Nk;
  const int kc = live ? key : Nk - 1;

  Q += b * sq + head * DH;
  Qt += b * sqt + (int64_t)head * DH * ldqt;
  K += b * sk + head * DH;
  V += b * sk + head * DH;
  dO += b * so + head * DH;
  dOt += b * sdot + (int64_t)head * DH * ldqt;
  dK += b * sk + head * DH;
  dV += b * sk + head * DH;
  LSE += ((int64_t)b * H + head) * Nq;
  Delta += ((int64_t)b * H + head) * Nq;

  bf16x8 kf[KS], vf[DO_DK ? KS : 1];
  load_row_frags<DH>(K + (int64_t)kc * ld, h, kf);
  if constexpr (DO_DK) load_row_frags<DH>(V + (int64_t)kc * ld, h, vf);
  f32x16 dk[DO_DK ? DT : 1], dv[DO_DV ? DT : 1];
#pragma unroll
  for (int i = 0; i < (DO_DK ? DT : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (DO_DV ? DT : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dv[i][e] = 0.f;

  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;
  const int roff = r * KROW + h * 16;
  int toff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) toff[c] = r * 128 + 8 * h + ((c ^ rswz) << 4);

  for (int q0 = 0; q0 < Nq; q0 += TB) {
    stage_rows<DH>(Q + (int64_t)q0 * ld, ld, sQ, t, wave);
    if constexpr (DO_DK) {
      stage_rows<DH>(dO + (int64_t)q0 * ldo, ldo, sG, t, wave);
      stage_cols<DH>(Qt + q0, ldqt, sQt, wave, rsub, cs);
    }
    if constexpr (DO_DV) stage_cols<DH>(dOt + q0, ldqt, sGt, wave, rsub, cs);
    if (t < TB) sL[t] = LSE[q0 + t];
    else if (t < 2 * TB) sD[t - TB] = Delta[q0 + t - TB];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 s, dp;
      // S = Q K^T and dP = dO V^T for 32 queries: queries on the accumulator rows, the key on the lane
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = *(const bf16x8*)(sQ + qt * 32 * KROW + roff + ks * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ks], ks == 0 ? zero : s, 0, 0, 0);
      }
      if constexpr (DO_DK) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 a = *(const bf16x8*)(sG + qt * 32 * KROW + roff + ks * 32);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, vf[ks], ks == 0 ? zero : dp, 0, 0, 0);
        }
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *(const f32x4*)(sL + 32 * qt + 8 * g + 4 * h);
        const f32x4 d4 = *(const f32x4*)(sD + 32 * qt + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = 4 * g + j;
          const float p = __builtin_amdgcn_exp2f(s[e] * scale_log2e - l4[j]);
          if constexpr (DO_DV) pf[e >> 3][e & 7] = (__bf16)p;
          if constexpr (DO_DK) dsf[e >> 3][e & 7] = (__bf16)(p * (dp[e] - d4[j]));
        }
      }
      // dV^T += dO^T P   and   dK^T += Q^T dS
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int c1 = 4 * qt + 2 * ss;
          if constexpr (DO_DV) {
            const bf16x8 g = frag_t(sGt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, pf[ss], dv[dt], 0, 0, 0);
          }
          if constexpr (DO_DK) {
            const bf16x8 a = frag_t(sQt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[ss], dk[dt], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }
  if (live) {
    if constexpr (DO_DK) store_rows<DH>(dK + (int64_t)key * ld, dk, scale, h);
    if constexpr (DO_DV) store_rows<DH>(dV + (int64_t)key * ld, dv, 1.0f, h);
  }
}

template <int DH>
void launch_bwd(const void* q, const void* qt, const void* k, const void* kt, const void* v, const void* o,
                const void* dout, const void* dout_t, const float* lse, float* delta, void* dq, void* dk, void* dv,
                int B, int H, int Nq, int Nk, int ld, int ld_o, int ld_t_q, int ld_t_k, int64_t sq, int64_t sk,
                int64_t so, int64_t sqt, int64_t sdot, int64_t skt, float scale, hipStream_t s) {
  const float sl2 = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(flash_bwd_dq_kernel<DH>, dim3((Nq + RB - 1) / RB, H, B), dim3(256), 0, s, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)kt, (const bf16_t*)dout, (const bf16_t*)o, lse,
                     (bf16_t*)dq, delta, Nq, Nk, ld, ld_o, ld_t_k, sq, sk, so, skt, scale, sl2);
#define DKV(WANT)                                                                                                       \
  hipLaunchKernelGGL((flash_bwd_dkv_kernel<DH, WANT>), dim3((Nk + RB - 1) / RB, H, B), dim3(256), 0, s, (const bf16_t*)q, \
                     (const bf16_t*)qt, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, (const bf16_t*)dout_t,  \
                     lse, delta, (bf16_t*)dk, (bf16_t*)dv, Nq, Nk, ld, ld_o, ld_t_q, sq, sqt, sk, so, sdot, scale, sl2)
  if constexpr (DH == 80) { DKV(1); DKV(2); }   // measured: two lighter walks beat one at a single wave per SIMD
  else DKV(3);
#undef DKV
}

}  // namespace

extern "C" int lbmk_flash_attn_bwd(const void* q, const void* qt, const void* k, const void* kt,
                                   const void* v, const void* o, const void* dout, const void* dout_t,
                                   const float* lse, float* delta, void* dq, void* dk, void* dv, int B,
                                   int H, int Nq, int Nk, int Dh, int ld, int ld_o, int ld_t_q, int ld_t_k,
                                   int64_t sq, int64_t sk, int64_t so, int64_t sqt, int64_t sdot, int64_t skt,
                                   float scale, void* stream) {
  LBMK_REQUIRE(q && qt && k && kt && v && o && dout && dout_t && lse && delta && dq && dk && dv,
               "flash_attn_bwd: null pointer");
  LBMK_REQUIRE(Dh == 64 || Dh == 40 || Dh == 80 || Dh == 160, "flash_attn_bwd: head_dim %d not supported (40, 64, 80, 160)", Dh);
  LBMK_REQUIRE(B > 0 && H > 0 && B <= 65535 && H <= 65535, "flash_attn_bwd: sizes");
  LBMK_REQUIRE(Nq > 0 && Nk > 0 && Nq % TB == 0 && Nk % TB == 0,
               "flash_attn_bwd: token counts (%d, %d) must be multiples of %d", Nq, Nk, TB);
  LBMK_REQUIRE(ld % 8 == 0 && ld >= H * Dh && ld_o % 8 == 0 && ld_o >= H * Dh && ld_t_q % 8 == 0 && ld_t_q >= Nq &&
                   ld_t_k % 8 == 0 && ld_t_k >= Nk, "flash_attn_bwd: leading dims");
  LBMK_REQUIRE(sq % 8 == 0 && sk % 8 == 0 && so % 8 == 0 && sqt % 8 == 0 && sdot % 8 == 0 && skt % 8 == 0 &&
                   sq >= (int64_t)Nq * ld && sk >= (int64_t)Nk * ld && so >= (int64_t)Nq * ld_o &&
                   sqt >= (int64_t)H * Dh * ld_t_q && sdot >= (int64_t)H * Dh * ld_t_q && skt >= (int64_t)H * Dh * ld_t_k,
               "flash_attn_bwd: batch strides");
  LBMK_REQUIRE(lbmk_aligned(q, 16) && lbmk_aligned(qt, 16) && lbmk_aligned(k, 16) && lbmk_aligned(kt, 16) &&
                   lbmk_aligned(v, 16) && lbmk_aligned(o, 16) && lbmk_aligned(dout, 16) &&
                   lbmk_aligned(dout_t, 16) && lbmk_aligned(dq, 8) && lbmk_aligned(dk, 8) && lbmk_aligned(dv, 8),
               "flash_attn_bwd: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
#define BWD(DH) launch_bwd<DH>(q, qt, k, kt, v, o, dout, dout_t, lse, delta, dq, dk, dv, B, H, Nq, Nk, ld, ld_o, ld_t_q, ld_t_k, \
                               sq, sk, so, sqt, sdot, skt, scale, s)
  switch (Dh) {
    case 40: BWD(40); break;
    case 64: BWD(64); break;
    case 80: BWD(80); break;
    case 160: BWD(160); break;
  }
#undef BWD
  return lbmk_check_launch("flash_attn_bwd");
}
