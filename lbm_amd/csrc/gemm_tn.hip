// Weight-gradient GEMM on untransposed operands:  out[n][k] = sum_m A[m][n] * B[m][k]
// (A = dY [M][N], B = X [M][K], both row-major bf16: the reduction index m is the ROW of both).
//
// Both tiles are staged in their natural layout ([m][128 columns], 256-byte rows) with LDS-DMA and read
// back with ds_read_b64_tr_b16, gfx950's transposing LDS read: per 16-lane group it takes a block of 4 rows
// x 16 columns and hands lane i column i of the 4 rows, which is exactly half of a lane's 8-deep k slice of
// a v_mfma_f32_16x16x32_bf16 operand.  So no transposed copy of dY or X is ever written to memory.
// The 16-byte slots of a row are XOR-swizzled (slot ^ (((row & 3) << 2) | ((row >> 2) & 3))) on the source
// side of the DMA and on the read address, which keeps both the DMA image lane-linear and the transposed
// reads free of bank conflicts.
// Workgroup = 4 waves = a 128 (n) x 128 (k) output tile, 64 x 64 per wave; 64 rows of m per stage, two
// stage buffers (64 KB), one barrier per stage.  blockIdx.z splits the m range; with more than one split
// the kernel writes f32 partials [split][N][K] that lbmk_sum_partials adds in a fixed order.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int TN_BN = 128, TN_BK = 128, TN_MS = 64;      // tile columns of A / of B, rows of m per stage
constexpr int TN_TILE = TN_MS * 256;                      // bytes of one operand's stage tile
constexpr unsigned TN_OOB = 0x80000000u;

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// 8-deep k slice (rows r0 .. r0+7 of the stage tile) of column `16 * f + (lane & 15)`: two transposing reads
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int r0, int f, int l) {
  const int q = l >> 2, p = l & 3;
  const int ra = r0 + q, rb = r0 + 4 + q;
  const int ch = 2 * f + (p >> 1);
  const char* pa = tile + 256 * ra + 16 * (ch ^ tn_swz(ra)) + 8 * (p & 1);
  const char* pb = tile + 256 * rb + 16 * (ch ^ tn_swz(rb)) + 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         void* __restrict__ out, int M, int N, int K, int lda, int ldb,
                                                         int ldo, int m_chunk, int tiles_k) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TN_TILE];   // [buffer][A tile | B tile]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_n * tiles_k;
  const int n0 = tile_n * TN_BN, k0 = tile_k * TN_BK;
  const int z = blockIdx.z;
  const int m_begin = z * m_chunk, m_end = min(m_begin + m_chunk, M);

  // staging: thread -> (row, slot) of four 16-byte chunks per operand; the source chunk is the swizzled one
  const int slot = lane & 15, rsub = lane >> 4;
  unsigned a_off[4], b_off[4];
  int srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + wave * 4 + rsub;
    const int ch = slot ^ tn_swz(row);
    srow[i] = row;
    a_off[i] = (n0 + ch * 8 < N) ? (unsigned)(row * lda + n0 + ch * 8) * 2u : TN_OOB;
    b_off[i] = (k0 + ch * 8 < K) ? (unsigned)(row * ldb + k0 + ch * 8) * 2u : TN_OOB;
  }
  // descriptors cover rows [m_begin, m_end) of each operand: rows past the end read zeros
  const bf16_t* a_base = A + (int64_t)m_begin * lda;
  const bf16_t* b_base = B + (int64_t)m_begin * ldb;
  const int64_t a_span = ((int64_t)(m_end - m_begin - 1) * lda + N) * 2, b_span = ((int64_t)(m_end - m_begin - 1) * ldb + K) * 2;
  const int a_bytes = (int)(a_span > 0x7fffffff ? 0x7fffffff : a_span), b_bytes = (int)(b_span > 0x7fffffff ? 0x7fffffff : b_span);

  auto stage = [&](int ms, char* buf) __attribute__((always_inline)) {
    const unsigned so_a = (unsigned)(ms * TN_MS) * (unsigned)lda * 2u, so_b = (unsigned)(ms * TN_MS) * (unsigned)ldb * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool row_ok = m_begin + ms * TN_MS + srow[i] < m_end;
      buffer_lds16(a_base, a_bytes, (row_ok && a_off[i] != TN_OOB) ? a_off[i] + so_a : TN_OOB, 0, buf + (i * 256 + wave * 64) * 16);
      buffer_lds16(b_base, b_bytes, (row_ok && b_off[i] != TN_OOB) ? b_off[i] + so_b : TN_OOB, 0,
                   buf + TN_TILE + (i * 256 + wave * 64) * 16);
    }
  };

  f32x4 acc[4][4];   // [k fragment][n fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l = lane & 15, g = lane >> 4;
  auto compute = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g;
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = tn_frag(buf, r0, wn * 4 + j, l);            // dY: columns n
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = tn_frag(buf + TN_TILE, r0, wk * 4 + i, l);  // X: columns k
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nms = (m_end - m_begin + TN_MS - 1) / TN_MS;
  stage(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ms = 0; ms < nms; ++ms) {
    if (ms + 1 < nms) stage(ms + 1, smem + ((ms + 1) & 1) * 2 * TN_TILE);
    compute(smem + (ms & 1) * 2 * TN_TILE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // lane holds out[n = n0 + 64 wn + 16 j + l][k = k0 + 64 wk + 16 i + 4 g .. +3]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + 16 * j + l;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + wk * 64 + 16 * i + 4 * g;
      if (k >= K) continue;                   // K % 8 == 0: a 4-column group is inside or outside as a whole
      const int64_t at = ((int64_t)z * N + n) * ldo + k;
      if (OUT_F32) *(f32x4*)((float*)out + at) = acc[i][j];
      else {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = f2bf(acc[i][j][e]);
        *(u16x4*)((bf16_t*)out + at) = o;
      }
    }
  }
}

}  // namespace

extern "C" int lbmk_gemm_tn(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo,
                            int splits, int out_f32, void* stream) {
  LBMK_REQUIRE(a && b && out, "gemm_tn: null pointer");
  LBMK_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "gemm_tn: M=%d N=%d K=%d (N, K multiples of 8)", M, N, K);
  LBMK_REQUIRE(lda >= N && ldb >= K && ldo >= K && lda % 8 == 0 && ldb % 8 == 0 && ldo % 4 == 0, "gemm_tn: leading dims");
  LBMK_REQUIRE(splits >= 1 && splits <= 65535 && (splits == 1 || out_f32), "gemm_tn: %d splits need f32 partials", splits);
  LBMK_REQUIRE(lbmk_aligned(a, 16) && lbmk_aligned(b, 16) && lbmk_aligned(out, out_f32 ? 16 : 8), "gemm_tn: misaligned");
  int m_chunk = (M + splits - 1) / splits;
  m_chunk = (m_chunk + TN_MS - 1) / TN_MS * TN_MS;
  // 32-bit byte offsets inside one split's window of each operand
  LBMK_REQUIRE((int64_t)(m_chunk + TN_MS) * lda * 2 < (1LL << 31) && (int64_t)(m_chunk + TN_MS) * ldb * 2 < (1LL << 31),
               "gemm_tn: a split of %d rows overflows 32-bit offsets; use more splits", m_chunk);
  LBMK_REQUIRE((int64_t)(splits - 1) * m_chunk < M, "gemm_tn: %d splits of %d rows leave an empty split", splits, m_chunk);
  const int tiles_n = (N + TN_BN - 1) / TN_BN, tiles_k = (K + TN_BK - 1) / TN_BK;
  dim3 grid((unsigned)(tiles_n * tiles_k), 1, (unsigned)splits);
  if (out_f32)
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, out,
                       M, N, K, lda, ldb, ldo, m_chunk, tiles_k);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, out,
                       M, N, K, lda, ldb, ldo, m_chunk, tiles_k);
  return lbmk_check_launch("gemm_tn");
}
