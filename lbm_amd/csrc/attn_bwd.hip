// Flash-style attention backward for head_dim 64 on v_mfma_f32_32x32x16_bf16 (gfx950).
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

namespace {

constexpr int D = 64;
constexpr int RB = 128;   // rows (queries or keys) owned by a workgroup
constexpr int TB = 64;    // tokens per tile of the walked operand

// stage a 64-row x 128-byte image: source row stride ld (elements), 16-byte slots XOR-swizzled
__device__ __forceinline__ void stage_tile(const bf16_t* src, int ld, char* dst, int wave, int rsub, int cs) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    glds16(src + (int64_t)(rsub + 32 * i) * ld + cs * 8, dst + (wave * 8 + 32 * i) * 128);
}

__device__ __forceinline__ bf16x8 frag_t(const char* img, int off_lo, int off_hi) {
  const uint2 lo = *(const uint2*)(img + off_lo);
  const uint2 hi = *(const uint2*)(img + off_hi);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

__device__ __forceinline__ void store_rows(bf16_t* op, const f32x16 (&acc)[2], float mul, int h) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(acc[dt][4 * g + e] * mul);
      *(u16x4*)(op + 32 * dt + 8 * g + 4 * h) = ov;
    }
}

__global__ __launch_bounds__(256, 2) void flash_bwd_dq_d64_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ Kt, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
    const float* __restrict__ LSE, bf16_t* __restrict__ dQ, float* __restrict__ Delta, int Nq, int Nk,
    int ldq, int ldk, int ldv, int ldkt, int lddo, int ldo, int lddq, int64_t sq, int64_t sk, int64_t sv,
    int64_t skt, int64_t sdo, int64_t so, int64_t sdq, float scale, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char smem[3 * TB * 128];
  char* sK = smem;                  // [key][64 d]
  char* sV = smem + TB * 128;       // [key][64 d]
  char* sKt = smem + 2 * TB * 128;  // [d][64 keys]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int q = blockIdx.x * RB + wave * 32 + r;

  Q += b * sq + head * D;
  K += b * sk + head * D;
  V += b * sv + head * D;
  Kt += b * skt + (int64_t)head * D * ldkt;
  dO += b * sdo + head * D;
  O += b * so + head * D;
  dQ += b * sdq + head * D;
  const int64_t stat = ((int64_t)b * H + head) * Nq + q;

  // B-operand fragments of this lane's query row: lane (r,h) holds X[q][16ks + 8h .. +8]
  bf16x8 qf[4], dof[4];
  float delta = 0.f;
  {
    const bf16_t* qp = Q + (int64_t)q * ldq + 8 * h;
    const bf16_t* gp = dO + (int64_t)q * lddo + 8 * h;
    const bf16_t* op = O + (int64_t)q * ldo + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + 16 * ks);
      dof[ks] = *(const bf16x8*)(gp + 16 * ks);
      const bf16x8 of = *(const bf16x8*)(op + 16 * ks);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += (float)dof[ks][e] * (float)of[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    if (h == 0) Delta[stat] = delta;
  }
  const float lse = LSE[stat];

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;
  int roff[4], toff[8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = r * 128 + (((2 * ks + h) ^ rswz) << 4);
#pragma unroll
  for (int c = 0; c < 8; ++c) toff[c] = r * 128 + 8 * h + ((c ^ rswz) << 4);

  for (int kv0 = 0; kv0 < Nk; kv0 += TB) {
    stage_tile(K + (int64_t)kv0 * ldk, ldk, sK, wave, rsub, cs);
    stage_tile(V + (int64_t)kv0 * ldv, ldv, sV, wave, rsub, cs);
    stage_tile(Kt + kv0, ldkt, sKt, wave, rsub, cs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 s, dp;
      // S^T = K Q^T and dP^T = V dO^T for 32 keys: keys on the accumulator rows
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(sK + kt * 32 * 128 + roff[ks]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero : s, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 vf = *(const bf16x8*)(sV + kt * 32 * 128 + roff[ks]);
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
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int c1 = 4 * kt + 2 * ss;
          const bf16x8 kf = frag_t(sKt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, dsf[ss], acc[dt], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  store_rows(dQ + (int64_t)q * lddq, acc, scale, h);
}

__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_d64_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Qt, const bf16_t* __restrict__ K,
    const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOt,
    const float* __restrict__ LSE, const float* __restrict__ Delta, bf16_t* __restrict__ dK,
    bf16_t* __restrict__ dV, int Nq, int Nk, int ldq, int ldqt, int ldk, int ldv, int lddo, int lddot,
    int lddk, int lddv, int64_t sq, int64_t sqt, int64_t sk, int64_t sv, int64_t sdo, int64_t sdot,
    int64_t sdk, int64_t sdv, float scale, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TB * 128 + 2 * TB * 4];
  char* sQ = smem;                    // [query][64 d]
  char* sG = smem + TB * 128;         // dO, [query][64 d]
  char* sQt = smem + 2 * TB * 128;    // [d][64 queries]
  char* sGt = smem + 3 * TB * 128;    // dO^T, [d][64 queries]
  float* sL = (float*)(smem + 4 * TB * 128);   // log-sum-exp of the 64 queries, then delta
  float* sD = sL + TB;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int key = blockIdx.x * RB + wave * 32 + r;

  Q += b * sq + head * D;
  Qt += b * sqt + (int64_t)head * D * ldqt;
  K += b * sk + head * D;
  V += b * sv + head * D;
  dO += b * sdo + head * D;
  dOt += b * sdot + (int64_t)head * D * lddot;
  dK += b * sdk + head * D;
  dV += b * sdv + head * D;
  LSE += ((int64_t)b * H + head) * Nq;
  Delta += ((int64_t)b * H + head) * Nq;

  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = K + (int64_t)key * ldk + 8 * h;
    const bf16_t* vp = V + (int64_t)key * ldv + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *(const bf16x8*)(kp + 16 * ks);
      vf[ks] = *(const bf16x8*)(vp + 16 * ks);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; }

  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;
  int roff[4], toff[8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) roff[ks] = r * 128 + (((2 * ks + h) ^ rswz) << 4);
#pragma unroll
  for (int c = 0; c < 8; ++c) toff[c] = r * 128 + 8 * h + ((c ^ rswz) << 4);

  for (int q0 = 0; q0 < Nq; q0 += TB) {
    stage_tile(Q + (int64_t)q0 * ldq, ldq, sQ, wave, rsub, cs);
    stage_tile(dO + (int64_t)q0 * lddo, lddo, sG, wave, rsub, cs);
    stage_tile(Qt + q0, ldqt, sQt, wave, rsub, cs);
    stage_tile(dOt + q0, lddot, sGt, wave, rsub, cs);
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
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *(const bf16x8*)(sQ + qt * 32 * 128 + roff[ks]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ks], ks == 0 ? zero : s, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *(const bf16x8*)(sG + qt * 32 * 128 + roff[ks]);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, vf[ks], ks == 0 ? zero : dp, 0, 0, 0);
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
          pf[e >> 3][e & 7] = (__bf16)p;
          dsf[e >> 3][e & 7] = (__bf16)(p * (dp[e] - d4[j]));
        }
      }
      // dV^T += dO^T P   and   dK^T += Q^T dS
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int c1 = 4 * qt + 2 * ss;
          const bf16x8 g = frag_t(sGt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, pf[ss], dv[dt], 0, 0, 0);
          const bf16x8 a = frag_t(sQt + dt * 32 * 128, toff[c1], toff[c1 + 1]);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[ss], dk[dt], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  store_rows(dK + (int64_t)key * lddk, dk, scale, h);
  store_rows(dV + (int64_t)key * lddv, dv, 1.0f, h);
}

}  // namespace

extern "C" int lbmk_flash_attn_bwd(const void* q, const void* qt, const void* k, const void* kt,
                                   const void* v, const void* o, const void* dout, const void* dout_t,
                                   const float* lse, float* delta, void* dq, void* dk, void* dv, int B,
                                   int H, int Nq, int Nk, int Dh, int ld, int ld_t_q, int ld_t_k,
                                   int64_t sq, int64_t sk, float scale, void* stream) {
  LBMK_REQUIRE(q && qt && k && kt && v && o && dout && dout_t && lse && delta && dq && dk && dv,
               "flash_attn_bwd: null pointer");
  LBMK_REQUIRE(Dh == 64, "flash_attn_bwd: head_dim %d not supported (64)", Dh);
  LBMK_REQUIRE(B > 0 && H > 0 && B <= 65535 && H <= 65535, "flash_attn_bwd: sizes");
  LBMK_REQUIRE(Nq > 0 && Nk > 0 && Nq % RB == 0 && Nk % RB == 0,
               "flash_attn_bwd: token counts (%d, %d) must be multiples of %d", Nq, Nk, RB);
  LBMK_REQUIRE(ld % 8 == 0 && ld >= H * Dh && ld_t_q % 8 == 0 && ld_t_q >= Nq && ld_t_k % 8 == 0 && ld_t_k >= Nk,
               "flash_attn_bwd: leading dims");
  LBMK_REQUIRE(sq % 8 == 0 && sk % 8 == 0 && sq >= (int64_t)Nq * ld && sk >= (int64_t)Nk * ld,
               "flash_attn_bwd: batch strides");
  LBMK_REQUIRE(lbmk_aligned(q, 16) && lbmk_aligned(qt, 16) && lbmk_aligned(k, 16) && lbmk_aligned(kt, 16) &&
                   lbmk_aligned(v, 16) && lbmk_aligned(o, 16) && lbmk_aligned(dout, 16) &&
                   lbmk_aligned(dout_t, 16) && lbmk_aligned(dq, 8) && lbmk_aligned(dk, 8) && lbmk_aligned(dv, 8),
               "flash_attn_bwd: misaligned pointer");
  hipStream_t s = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  // channel-major copies are [B][H*Dh][ld_t]
  const int64_t sqt = (int64_t)H * Dh * ld_t_q, skt = (int64_t)H * Dh * ld_t_k;
  hipLaunchKernelGGL(flash_bwd_dq_d64_kernel, dim3(Nq / RB, H, B), dim3(256), 0, s, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)kt, (const bf16_t*)dout,
                     (const bf16_t*)o, lse, (bf16_t*)dq, delta, Nq, Nk, ld, ld, ld, ld_t_k, ld, ld, ld, sq, sk,
                     sk, skt, sq, sq, sq, scale, sl2);
  hipLaunchKernelGGL(flash_bwd_dkv_d64_kernel, dim3(Nk / RB, H, B), dim3(256), 0, s, (const bf16_t*)q,
                     (const bf16_t*)qt, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout,
                     (const bf16_t*)dout_t, lse, delta, (bf16_t*)dk, (bf16_t*)dv, Nq, Nk, ld, ld_t_q, ld, ld, ld,
                     ld_t_q, ld, ld, sq, sqt, sk, sk, sq, sqt, sk, sk, scale, sl2);
  return lbmk_check_launch("flash_attn_bwd");
}
