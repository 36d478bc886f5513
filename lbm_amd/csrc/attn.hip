// Flash-style attention forward on v_mfma_f32_32x32x16_bf16 (gfx950): a tuned head_dim-64 kernel and a
// generic one for head dims 8 ... 160 (multiples of 8, zero-padded to the MFMA shapes), plus the row
// softmax used by the unfused (head_dim 512) VAE attention.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows.
// Per 64-key tile:  S^T = K Q^T  (keys on the accumulator rows, the query row on the lane),
// so the online-softmax statistics of a query row live in one lane pair (l, l^32) and the
// exponentiated tile is — after a bf16 pack — directly the B operand of  O^T += V^T P^T
// (an accumulator tile used as the next MFMA's operand, no LDS round trip for P).
// V is consumed transposed ([channel][key]), which the projection GEMM writes for free.
#include "common.h"

__device__ __attribute__((aligned(16))) uint32_t attn_zero_page[8];

namespace {

constexpr int D = 64;
constexpr int QB = 128;   // query rows per workgroup
constexpr int KB = 64;    // keys per tile

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The softmax arithmetic, not the MFMA pipe, bounds this kernel (per score: scale-and-shift, exp2, sum,
// bf16 pack, running max on the vector ALU against 16 MFMA cycles), so the loop keeps vector work
// minimal: scalar fma / add pairs (packed f32 instructions issue slower beside MFMAs: +2 % measured with the
// file built -fno-slp-vectorize), LDS read offsets hoisted out of the loop, and the rescale of the output
// accumulator only when some row's running maximum actually moved.  116 registers; 3 and 4 waves per SIMD measure
// the same.
//
// FAST (inference; no log-sum-exp wanted): two vector instructions per score fewer.  (1) Q is multiplied by
// scale * log2(e) once, when its fragments are loaded (one more bf16 rounding of an operand that was rounded to bf16
// when it was stored), so the scores leave the MFMA in the exp2 domain; (2) the running maximum is subtracted INSIDE
// the MFMA: the first product of a score tile takes -m (16 registers, rewritten only when m changes) as its C operand,
// so p = exp2(s') needs no fma.  m is allowed to lag: it is raised (output and row sum rescaled, the tile's scores
// shifted) only when some row's tile maximum exceeds it by more than 2^10, so p <= 1024 and the rescale branch runs a
// handful of times per row; the softmax is unchanged up to f32 rounding (every term and the row sum carry the same
// 2^-m).  Why this and not an overlap of the softmax with the MFMAs: on this chip a vector instruction beside MFMAs is
// not free -- scripts/micro/mfma_filler.hip measures 32 cycles per v_mfma_f32_32x32x16_bf16 with up to 3 v_fma_f32
// (or 2 v_exp_f32) behind it, and +1.9 cycles per further v_fma_f32 (+6.1 per v_exp_f32), for 1, 2 or 3 waves per SIMD
// alike -- so the loop's time is its MFMA time plus most of its vector time, and only fewer vector instructions help.
template <bool FAST>
__global__ __launch_bounds__(256, 3) void flash_fwd_d64_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
    bf16_t* __restrict__ O, int Nq, int Nk, int ldq, int ldk, int ldvt, int ldo, int64_t sq,
    int64_t sk, int64_t svt, int64_t so, float scale_log2e, float* __restrict__ LSE) {
  // two stage buffers of (K tile, V^T tile): the loads of tile i+1 are issued right after the barrier that
  // publishes tile i and land under its MFMAs / softmax; one barrier per tile
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KB * 128];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * QB + wave * 32;

  Q += b * sq + head * D;
  K += b * sk + head * D;
  Vt += b * svt + (int64_t)head * D * ldvt;
  O += b * so + head * D;

  // Q fragments: B operand, lane (r,h) holds Q[q0+r][16ks + 8h .. +8]
  bf16x8 qf[4];
  {
    const int qr = min(q0 + r, Nq - 1);
    const bf16_t* qp = Q + (int64_t)qr * ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + 16 * ks);
      if (FAST) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)((float)qf[ks][e] * scale_log2e);
      }
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // FAST: -c_m in every element, the C operand of a score tile's first product (c_m = 0 until the first tile sets it)
  f32x16 negm;
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[e] = 0.f;
  float c_m = 0.f;
  constexpr float LAG = 10.f;      // m may lag a row's true maximum by 2^LAG

  // staging coordinates: two rows per thread for each of the K and V^T images
  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;
  // loop-invariant LDS read offsets of this lane's fragments
  int koff[4], voff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ rswz) << 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) voff[c] = r * 128 + (((2 * c + h) ^ rswz) << 4);   // 8 keys 16 c + 8 h .. of V^T row r

  auto stage = [&](int kv0, char* buf) __attribute__((always_inline)) {
    char* dK = buf;               // [key][64 d]    128-byte rows, 16-byte slots XOR-swizzled
    char* dV = buf + KB * 128;    // [d][64 keys]   same image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rsub + 32 * i;
      const int key = kv0 + row;
      const bf16_t* src = key < Nk ? K + (int64_t)key * ldk + cs * 8 : (const bf16_t*)attn_zero_page;
      glds16(src, dK + (wave * 8 + 32 * i) * 128);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rsub + 32 * i;   // channel d
      const int key = kv0 + cs * 8;
      const bf16_t* src = key < Nk ? Vt + (int64_t)row * ldvt + key : (const bf16_t*)attn_zero_page;
      glds16(src, dV + (wave * 8 + 32 * i) * 128);
    }
  };
  stage(0, smem);
  int cur = 0;
  for (int kv0 = 0; kv0 < Nk; kv0 += KB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // tile kv0 is in place, and every wave is done reading the other buffer (tile kv0 - KB)
    const char* sK = smem + cur * (2 * KB * 128);
    const char* sV = sK + KB * 128;
    cur ^= 1;
    if (kv0 + KB < Nk) stage(kv0 + KB, smem + cur * (2 * KB * 128));

    // ---- S^T = K Q^T : two 32-key sub-tiles -------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(sK + kt * 32 * 128 + koff[ks]);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? (FAST ? negm : zero) : s[kt], 0, 0, 0);
      }
    }
    if (kv0 + KB > Nk) {   // mask the keys past the end (wave-uniform branch)
      // the empty asm keeps this a branch: without it the compiler turns the body into 32 compare / select pairs
      // that run for EVERY tile (a quarter of the loop's vector instructions)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kv0 + 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= Nk) s[kt][e] = -INFINITY;
        }
    }
    // ---- online softmax (row = lane pair (l, l^32)) -----------------------------------
    float ps0 = 0.f, ps1 = 0.f;
    bf16x8 pf[2][2];
    if (FAST) {
      // s holds s' = scores (exp2 domain) - c_m.  Tile maximum as a tree of three-input maxima (v_max3_f32).
      float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[1][0]);
#pragma unroll
      for (int e = 2; e < 16; e += 2) mx = fmaxf(fmaxf(mx, s[0][e]), s[0][e + 1]);
#pragma unroll
      for (int e = 1; e < 15; e += 2) mx = fmaxf(fmaxf(mx, s[1][e]), s[1][e + 1]);
      mx = fmaxf(mx, s[1][15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (__builtin_amdgcn_ballot_w64(mx + c_m - LAG > m_run) != 0) {   // some row outgrew m by more than 2^LAG (or first tile)
        const float m_new = fmaxf(m_run, mx + c_m);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        const float d = m_new - c_m;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kt][e] -= d;
        m_run = c_m = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[e] = -m_new;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[kt][e]);
          const float p1 = __builtin_amdgcn_exp2f(s[kt][e + 1]);
          ps0 += p0;
          ps1 += p1;
          pf[kt][e >> 3][e & 7] = (__bf16)p0;
          pf[kt][e >> 3][(e & 7) + 1] = (__bf16)p1;
        }
    } else {
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2e);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // some row's maximum moved: rescale
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const f32x2 a2 = {alpha, alpha};
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          f32x2 v = {o[i][e], o[i][e + 1]};
          v *= a2;
          o[i][e] = v.x;
          o[i][e + 1] = v.y;
        }
      m_run = m_new;
    }
    const float nm = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], scale_log2e, nm));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e + 1], scale_log2e, nm));
        ps0 += p0;
        ps1 += p1;
        pf[kt][e >> 3][e & 7] = (__bf16)p0;
        pf[kt][e >> 3][(e & 7) + 1] = (__bf16)p1;
      }
    }
    l_run += ps0 + ps1;
    // A lane pair (l, l + 32) holds the keys of a 16-key step interleaved in fours (l: 0-3 and 8-11, l + 32: 4-7 and
    // 12-15).  Swapping the second packed half of l with the first of l + 32 gives each lane 8 CONSECUTIVE keys
    // (l: 0-7, l + 32: 8-15), so its V^T fragment is one 16-byte LDS read instead of two 8-byte reads glued by moves.
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const u32x4 w = __builtin_bit_cast(u32x4, pf[kt][ss]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
        const u32x4 x = {s0[0], s1[0], s0[1], s1[1]};
        pf[kt][ss] = __builtin_bit_cast(bf16x8, x);
      }

    // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 vf = *(const bf16x8*)(sV + dt * 32 * 128 + voff[2 * kt + ss]);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][ss], o[dt], 0, 0, 0);
        }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < Nq) {
    // log2-domain log-sum-exp of the scaled scores, kept for the backward pass
    if (LSE && h == 0) LSE[((int64_t)b * gridDim.y + head) * Nq + q] = m_run + __builtin_amdgcn_logf(l_tot);
    bf16_t* op = O + (int64_t)q * ldo;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(o[dt][4 * g + e] * inv);
        *(u16x4*)(op + 32 * dt + 8 * g + 4 * h) = ov;
      }
  }
}

// Same algorithm for any head_dim that is a multiple of 8 (the SD1.5-style UNet has 40 / 80 / 160):
// the K image is [key][ceil16(D)] with the pad chunk read from the zero page, O^T has ceil(D/32) tiles.
// FAST: as in the head_dim-64 kernel (prescaled q, the lagging maximum subtracted as the score product's C operand).
template <int DH, bool FAST>
__global__ __launch_bounds__(256) void flash_fwd_gen_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
    bf16_t* __restrict__ O, int Nq, int Nk, int ldq, int ldk, int ldvt, int ldo, int64_t sq,
    int64_t sk, int64_t svt, int64_t so, float scale_log2e, float* __restrict__ LSE) {
  constexpr int KS = (DH + 15) / 16;     // 16-deep MFMA steps over the head dim
  constexpr int DT = (DH + 31) / 32;     // 32-row tiles of O^T
  constexpr int KROW = KS * 32;          // bytes per key row of the K image
  constexpr int KCHUNKS = KB * KS * 2;   // 16-byte chunks of one K tile
  constexpr int KI = (KCHUNKS + 255) / 256;
  constexpr int STAGE = KB * KROW + DT * 32 * 128;   // one (K tile, V^T tile) pair
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];   // two stage buffers, as in the head_dim-64 kernel

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * QB + wave * 32;

  Q += b * sq + head * DH;
  K += b * sk + head * DH;
  Vt += b * svt + (int64_t)head * DH * ldvt;
  O += b * so + head * DH;

  bf16x8 qf[KS];
  {
    const int qr = min(q0 + r, Nq - 1);
    const bf16_t* qp = Q + (int64_t)qr * ldq + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (16 * ks + 8 * h + 8 <= DH) {
        qf[ks] = *(const bf16x8*)(qp + 16 * ks);
        if (FAST) {
#pragma unroll
          for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)((float)qf[ks][e] * scale_log2e);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)0.f;
      }
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 negm;
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[e] = 0.f;
  float c_m = 0.f;
  constexpr float LAG = 10.f;

  const int slot = t & 7, rsub = t >> 3;
  const int cs = slot ^ ((rsub >> 1) & 7);
  const int rswz = (r >> 1) & 7;

  auto stage = [&](int kv0, char* buf) __attribute__((always_inline)) {
    char* dK = buf;
    char* dV = buf + KB * KROW;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int idx = i * 256 + t;
      if (idx < KCHUNKS) {                       // whole waves are in or out (KCHUNKS % 64 == 0)
        const int row = idx / (KS * 2), c = idx - row * (KS * 2);
        const int key = kv0 + row;
        const bf16_t* src = (key < Nk && c * 8 < DH) ? K + (int64_t)key * ldk + c * 8
                                                     : (const bf16_t*)attn_zero_page;
        glds16(src, dK + (i * 256 + wave * 64) * 16);
      }
    }
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int row = rsub + 32 * i;             // channel d
      const int key = kv0 + cs * 8;
      const bf16_t* src = (row < DH && key < Nk) ? Vt + (int64_t)row * ldvt + key
                                                 : (const bf16_t*)attn_zero_page;
      glds16(src, dV + (wave * 8 + 32 * i) * 128);
    }
  };
  stage(0, smem);
  int cur = 0;
  for (int kv0 = 0; kv0 < Nk; kv0 += KB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // tile kv0 is in place, and every wave is done reading the other buffer
    const char* sK = smem + cur * STAGE;
    const char* sV = sK + KB * KROW;
    cur ^= 1;
    if (kv0 + KB < Nk) stage(kv0 + KB, smem + cur * STAGE);

    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kt][e] = FAST ? negm[e] : 0.f;
      const char* krow = sK + (32 * kt + r) * KROW + h * 16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(krow + ks * 32);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    if (kv0 + KB > Nk) {
      asm volatile("" ::: "memory");   // keeps the tail mask a branch (see flash_fwd_d64_kernel)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kv0 + 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= Nk) s[kt][e] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
    bf16x8 pf[2][2];
    if (FAST) {
      if (__builtin_amdgcn_ballot_w64(mx + c_m - LAG > m_run) != 0) {   // some row outgrew m by more than 2^LAG (or first tile)
        const float m_new = fmaxf(m_run, mx + c_m);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        const float d = m_new - c_m;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kt][e] -= d;
        m_run = c_m = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[e] = -m_new;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pv = __builtin_amdgcn_exp2f(s[kt][e]);
          psum += pv;
          pf[kt][e >> 3][e & 7] = (__bf16)pv;
        }
    } else {
      const float m_new = fmaxf(m_run, mx * scale_log2e);
      if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // some row's maximum moved: rescale
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        m_run = m_new;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pv = __builtin_amdgcn_exp2f(s[kt][e] * scale_log2e - m_new);
          psum += pv;
          pf[kt][e >> 3][e & 7] = (__bf16)pv;
        }
    }
    l_run += psum;
    // lane pair exchange: 8 consecutive keys per lane and 16-key step (see the head_dim-64 kernel)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const u32x4 w = __builtin_bit_cast(u32x4, pf[kt][ss]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
        const u32x4 x = {s0[0], s1[0], s0[1], s1[1]};
        pf[kt][ss] = __builtin_bit_cast(bf16x8, x);
      }

#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const char* vrow = sV + (32 * dt + r) * 128;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 vf = *(const bf16x8*)(vrow + (((4 * kt + 2 * ss + h) ^ rswz) << 4));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][ss], o[dt], 0, 0, 0);
        }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < Nq) {
    // log2-domain log-sum-exp of the scaled scores, kept for the backward pass
    if (LSE && h == 0) LSE[((int64_t)b * gridDim.y + head) * Nq + q] = m_run + __builtin_amdgcn_logf(l_tot);
    bf16_t* op = O + (int64_t)q * ldo;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * dt + 8 * g + 4 * h;
        if (d0 < DH) {
          u16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(o[dt][4 * g + e] * inv);
          *(u16x4*)(op + d0) = ov;
        }
      }
  }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S,
                                                           bf16_t* __restrict__ P, int cols,
                                                           int64_t ld_in, int64_t ld_out) {
  __shared__ float red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* s = S + (int64_t)blockIdx.x * ld_in;
  bf16_t* p = P + (int64_t)blockIdx.x * ld_out;
  const int nv = cols >> 2;
  float mx = -INFINITY;
  for (int i = t; i < nv; i += 256) {
    const f32x4 v = *(const f32x4*)(s + 4 * i);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = t; i < nv; i += 256) {
    const f32x4 v = *(const f32x4*)(s + 4 * i);
    sum += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = t; i < nv; i += 256) {
    const f32x4 v = *(const f32x4*)(s + 4 * i);
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = f2bf(__expf(v[e] - mx) * inv);
    *(u16x4*)(p + 4 * i) = o;
  }
}

// Same softmax with the row held in registers (cols <= 4 * NT * NV): one read of the f32 scores instead of three.
template <int NV, int NT = 256>
__global__ __launch_bounds__(NT) void softmax_rows_reg_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                              int cols, int64_t ld_in, int64_t ld_out) {
  constexpr int NWV = NT / 64;
  __shared__ float red[2 * NWV];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* s = S + (int64_t)blockIdx.x * ld_in;
  bf16_t* p = P + (int64_t)blockIdx.x * ld_out;
  const int nv = cols >> 2;
  f32x4 v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = t + NT * k;
    v[k] = i < nv ? *(const f32x4*)(s + 4 * i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mx = fmaxf(fmaxf(mx, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[k][e] = __expf(v[k][e] - mx); sum += v[k][e]; }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[NWV + wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) tot += red[NWV + w];     // fixed order: run-to-run reproducible
  const float inv = 1.0f / tot;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = t + NT * k;
    if (i < nv) {
      u16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = f2bf(v[k][e] * inv);
      *(u16x4*)(p + 4 * i) = o;
    }
  }
}

}  // namespace

extern "C" int lbmk_flash_attn_fwd(const void* q, const void* k, const void* vt, void* o, int B,
                                   int H, int Nq, int Nk, int Dh, int ldq, int ldk, int ldvt,
                                   int ldo, int64_t sq, int64_t sk, int64_t svt, int64_t so,
                                   float scale, float* lse, void* stream) {
  LBMK_REQUIRE(q && k && vt && o, "flash_attn: null pointer");
  LBMK_REQUIRE(Dh == 64 || Dh == 40 || Dh == 80 || Dh == 160 || Dh == 8 || Dh == 16 || Dh == 32,
               "flash_attn: head_dim %d not supported (8, 16, 32, 40, 64, 80, 160)", Dh);
  LBMK_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && B <= 65535 && H <= 65535, "flash_attn: sizes");
  LBMK_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "flash_attn: leading dims");
  LBMK_REQUIRE(sq % 8 == 0 && sk % 8 == 0 && svt % 8 == 0 && so % 4 == 0, "flash_attn: batch strides");
  // V^T rows are read in 8-key chunks: the row stride must cover Nk rounded up to 8 (padding finite)
  LBMK_REQUIRE(ldq >= H * Dh && ldk >= H * Dh && ldo >= H * Dh && ldvt >= (Nk + 7) / 8 * 8,
               "flash_attn: leading dims too small");
  LBMK_REQUIRE(lbmk_aligned(q, 16) && lbmk_aligned(k, 16) && lbmk_aligned(vt, 16) && lbmk_aligned(o, 8),
               "flash_attn: misaligned pointer");
  dim3 grid((Nq + QB - 1) / QB, H, B);
  hipStream_t s = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  static const bool no_fast = getenv("LBMK_NO_FLASH_FAST") != nullptr;
  const bool fast = !lse && !no_fast;
#define FA_ARGS (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)o, Nq, Nk, ldq, ldk, ldvt, ldo, sq, sk, svt, so, sl2, lse
  switch (Dh) {
    case 64:
      // the log-sum-exp rows feed the backward kernels, which recompute the scores from the unscaled q: keep that
      // pass's arithmetic; LBMK_NO_FLASH_FAST=1 does the same for inference (same-box A/B runs)
      if (!fast) hipLaunchKernelGGL(flash_fwd_d64_kernel<false>, grid, dim3(256), 0, s, FA_ARGS);
      else hipLaunchKernelGGL(flash_fwd_d64_kernel<true>, grid, dim3(256), 0, s, FA_ARGS);
      break;
    case 8: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<8, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<8, false>), grid, dim3(256), 0, s, FA_ARGS); break;
    case 16: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<16, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<16, false>), grid, dim3(256), 0, s, FA_ARGS); break;
    case 32: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<32, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<32, false>), grid, dim3(256), 0, s, FA_ARGS); break;
    case 40: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<40, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<40, false>), grid, dim3(256), 0, s, FA_ARGS); break;
    case 80: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<80, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<80, false>), grid, dim3(256), 0, s, FA_ARGS); break;
    case 160: if (fast) hipLaunchKernelGGL((flash_fwd_gen_kernel<160, true>), grid, dim3(256), 0, s, FA_ARGS); else hipLaunchKernelGGL((flash_fwd_gen_kernel<160, false>), grid, dim3(256), 0, s, FA_ARGS); break;
  }
#undef FA_ARGS
  return lbmk_check_launch("flash_attn");
}

extern "C" int lbmk_softmax_rows(const float* s, void* p, int64_t rows, int cols, int64_t ld_in,
                                 int64_t ld_out, void* stream) {
  LBMK_REQUIRE(s && p, "softmax_rows: null pointer");
  LBMK_REQUIRE(rows > 0 && rows < (1LL << 31) && cols > 0 && cols % 4 == 0, "softmax_rows: rows/cols");
  LBMK_REQUIRE(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= cols && ld_out >= cols, "softmax_rows: leading dims");
  LBMK_REQUIRE(lbmk_aligned(s, 16) && lbmk_aligned(p, 8), "softmax_rows: misaligned");
#define SM_REG(NV) hipLaunchKernelGGL(softmax_rows_reg_kernel<NV>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, \
                                      (bf16_t*)p, cols, ld_in, ld_out)
  if (cols <= 1024) SM_REG(1);
  else if (cols <= 4096) SM_REG(4);
  else if (cols <= 16384) SM_REG(16);
  else if (cols <= 65536)    // 1024 threads x 16 float4: the 64k-token rows of the VAE attention at 2048^2
    hipLaunchKernelGGL((softmax_rows_reg_kernel<16, 1024>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, s,
                       (bf16_t*)p, cols, ld_in, ld_out);
  else
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s,
                       (bf16_t*)p, cols, ld_in, ld_out);
#undef SM_REG
  return lbmk_check_launch("softmax_rows");
}
