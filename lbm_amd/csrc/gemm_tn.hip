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

template <bool OUT_F32, int MS>
__global__ __launch_bounds__(256, MS == 64 ? 2 : 3) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         void* __restrict__ out, int M, int N, int K, int lda, int ldb,
                                                         int ldo, int m_chunk, int tiles_k) {
  __shared__ __attribute__((aligned(16))) char smem[4 * MS * 256];   // [buffer][A tile | B tile]
  constexpr int TILE = MS * 256;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_n * tiles_k;
  const int n0 = tile_n * TN_BN, k0 = tile_k * TN_BK;
  const int z = blockIdx.z;
  const int m_begin = z * m_chunk, m_end = min(m_begin + m_chunk, M);

  // staging: thread -> (row, slot) of four 16-byte chunks per operand; the source chunk is the swizzled one
  const int slot = lane & 15, rsub = lane >> 4;
  constexpr int NI = MS / 16;
  unsigned a_off[NI], b_off[NI];
  int srow[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
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
    const unsigned so_a = (unsigned)(ms * MS) * (unsigned)lda * 2u, so_b = (unsigned)(ms * MS) * (unsigned)ldb * 2u;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool row_ok = m_begin + ms * MS + srow[i] < m_end;
      buffer_lds16(a_base, a_bytes, (row_ok && a_off[i] != TN_OOB) ? a_off[i] + so_a : TN_OOB, 0, buf + (i * 256 + wave * 64) * 16);
      buffer_lds16(b_base, b_bytes, (row_ok && b_off[i] != TN_OOB) ? b_off[i] + so_b : TN_OOB, 0,
                   buf + TILE + (i * 256 + wave * 64) * 16);
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
    for (int ks = 0; ks < MS / 32; ++ks) {
      const int r0 = 32 * ks + 8 * g;
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = tn_frag(buf, r0, wn * 4 + j, l);            // dY: columns n
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = tn_frag(buf + TILE, r0, wk * 4 + i, l);  // X: columns k
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nms = (m_end - m_begin + MS - 1) / MS;
  stage(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ms = 0; ms < nms; ++ms) {
    if (ms + 1 < nms) stage(ms + 1, smem + ((ms + 1) & 1) * 2 * TILE);
    compute(smem + (ms & 1) * 2 * TILE);
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

// ---------------------------------------------------------------------------------------------------------------
// Input-gradient GEMM on the untransposed weight:  out[m][k] = sum_n A[m][n] * B[n][k]
// (A = dY [M][N] row-major: the reduction index n is contiguous, the usual A operand; B = W [N][K] row-major, the
// parameter itself: the reduction index is its ROW, so its tile is staged in the natural [n][128 columns] layout and
// read with the transposing LDS read, exactly like the operands of the weight-gradient kernel above).  No W^T copy.
// Workgroup = 4 waves = a 128 (m) x 128 (k) output tile, 64 x 64 per wave; 64 of n per stage, two stage buffers.
// blockIdx.z splits the n range (f32 partials [split][M][K], summed in a fixed order by lbmk_sum_partials).
constexpr int NN_BM = 128, NN_BK = 128, NN_NS = 64;

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         void* __restrict__ out, int M, int N, int K, int lda, int ldb,
                                                         int ldo, int n_chunk, int tiles_k) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TN_TILE];   // [buffer][A tile 128 x 128 B | B tile 64 x 256 B]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wk = wave & 1;
  const int tile_m = blockIdx.x / tiles_k, tile_k = blockIdx.x - tile_m * tiles_k;
  const int m0 = tile_m * NN_BM, k0 = tile_k * NN_BK;
  const int z = blockIdx.z;
  const int n_begin = z * n_chunk, n_end = min(n_begin + n_chunk, N);

  // A staging: chunk idx = i * 256 + t -> (row = idx >> 3, slot = idx & 7); LDS image lane-linear, the 16-byte slots of
  // a row XOR-swizzled on the source side and on the read address (conflict-free ds_read_b128 of 16 rows x one chunk)
  unsigned a_off[4];
  int a_col[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = i * 256 + t, row = idx >> 3, slot = idx & 7;
    const int ch = slot ^ ((row >> 1) & 7);
    a_col[i] = ch * 8;
    a_off[i] = (m0 + row < M) ? (unsigned)(row * lda + ch * 8) * 2u : TN_OOB;
  }
  // B staging as in gemm_tn_kernel: (row = reduction index, 16 slots of 8 columns)
  const int slot = lane & 15, rsub = lane >> 4;
  unsigned b_off[4];
  int srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + wave * 4 + rsub;
    const int ch = slot ^ tn_swz(row);
    srow[i] = row;
    b_off[i] = (k0 + ch * 8 < K) ? (unsigned)(row * ldb + k0 + ch * 8) * 2u : TN_OOB;
  }
  const bf16_t* a_base = A + (int64_t)m0 * lda + n_begin;
  const bf16_t* b_base = B + (int64_t)n_begin * ldb;
  const int rows_a = min(NN_BM, M - m0);
  const int64_t a_span = ((int64_t)(rows_a - 1) * lda + (n_end - n_begin)) * 2;
  const int64_t b_span = ((int64_t)(n_end - n_begin - 1) * ldb + K) * 2;
  const int a_bytes = (int)(a_span > 0x7fffffff ? 0x7fffffff : a_span), b_bytes = (int)(b_span > 0x7fffffff ? 0x7fffffff : b_span);

  auto stage = [&](int ns, char* buf) __attribute__((always_inline)) {
    const int nb = n_begin + ns * NN_NS;
    const unsigned so_a = (unsigned)(ns * NN_NS) * 2u, so_b = (unsigned)(ns * NN_NS) * (unsigned)ldb * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool col_ok = nb + a_col[i] < n_end;         // N % 8 == 0: a chunk is inside or outside as a whole
      buffer_lds16(a_base, a_bytes, (col_ok && a_off[i] != TN_OOB) ? a_off[i] + so_a : TN_OOB, 0, buf + (i * 256 + wave * 64) * 16);
      const bool row_ok = nb + srow[i] < n_end;
      buffer_lds16(b_base, b_bytes, (row_ok && b_off[i] != TN_OOB) ? b_off[i] + so_b : TN_OOB, 0,
                   buf + TN_TILE + (i * 256 + wave * 64) * 16);
    }
  };

  f32x4 acc[4][4];   // [k fragment][m fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l = lane & 15, g = lane >> 4;
  auto compute = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                     // dY: row m, reduction elements 32 ks + 8 g .. + 7
        const int r = wm * 64 + 16 * j + l;
        fa[j] = *(const bf16x8*)(buf + r * 128 + (((4 * ks + g) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = tn_frag(buf + TN_TILE, 32 * ks + 8 * g, wk * 4 + i, l);   // W: columns k
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nns = (n_end - n_begin + NN_NS - 1) / NN_NS;
  stage(0, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ns = 0; ns < nns; ++ns) {
    if (ns + 1 < nns) stage(ns + 1, smem + ((ns + 1) & 1) * 2 * TN_TILE);
    compute(smem + (ns & 1) * 2 * TN_TILE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // lane holds out[m = m0 + 64 wm + 16 j + l][k = k0 + 64 wk + 16 i + 4 g .. +3]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + 16 * j + l;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + wk * 64 + 16 * i + 4 * g;
      if (k >= K) continue;
      const int64_t at = ((int64_t)z * M + m) * ldo + k;
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

extern "C" int lbmk_gemm_nn(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo,
                            int splits, int out_f32, void* stream) {
  LBMK_REQUIRE(a && b && out, "gemm_nn: null pointer");
  LBMK_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "gemm_nn: M=%d N=%d K=%d (N, K multiples of 8)", M, N, K);
  LBMK_REQUIRE(lda >= N && ldb >= K && ldo >= K && lda % 8 == 0 && ldb % 8 == 0 && ldo % 4 == 0, "gemm_nn: leading dims");
  LBMK_REQUIRE(splits >= 1 && splits <= 65535 && (splits == 1 || out_f32), "gemm_nn: %d splits need f32 partials", splits);
  LBMK_REQUIRE(lbmk_aligned(a, 16) && lbmk_aligned(b, 16) && lbmk_aligned(out, out_f32 ? 16 : 8), "gemm_nn: misaligned");
  int n_chunk = (N + splits - 1) / splits;
  n_chunk = (n_chunk + NN_NS - 1) / NN_NS * NN_NS;
  LBMK_REQUIRE((int64_t)(splits - 1) * n_chunk < N, "gemm_nn: %d splits of %d leave an empty split", splits, n_chunk);
  // 32-bit byte offsets inside one tile's window of A (128 rows) and one split's window of B
  LBMK_REQUIRE((int64_t)(NN_BM + 1) * lda * 2 < (1LL << 31) && (int64_t)(n_chunk + NN_NS) * ldb * 2 < (1LL << 31),
               "gemm_nn: operand window overflows 32-bit offsets");
  const int64_t tiles_m = (M + NN_BM - 1) / NN_BM, tiles_k = (K + NN_BK - 1) / NN_BK;
  LBMK_REQUIRE(tiles_m * tiles_k < (1LL << 31), "gemm_nn: grid");
  dim3 grid((unsigned)(tiles_m * tiles_k), 1, (unsigned)splits);
  if (out_f32)
    hipLaunchKernelGGL(gemm_nn_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, out,
                       M, N, K, lda, ldb, ldo, n_chunk, (int)tiles_k);
  else
    hipLaunchKernelGGL(gemm_nn_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, out,
                       M, N, K, lda, ldb, ldo, n_chunk, (int)tiles_k);
  return lbmk_check_launch("gemm_nn");
}

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
  // 32-row stages (32 KB of LDS, three workgroups per CU) pay only when the grid holds that many: measured 83 -> 68 us on
  // the 10240 x 1280 feed-forward weight (800 tiles), 5-15 % slower than 64-row stages on grids of 100-400 tiles
  const bool ms32 = (int64_t)tiles_n * tiles_k * splits >= 768;
#define TN_LAUNCH(F32, MSV) hipLaunchKernelGGL((gemm_tn_kernel<F32, MSV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, \
                                               (const bf16_t*)b, out, M, N, K, lda, ldb, ldo, m_chunk, tiles_k)
  if (out_f32) { if (ms32) TN_LAUNCH(true, 32); else TN_LAUNCH(true, 64); }
  else { if (ms32) TN_LAUNCH(false, 32); else TN_LAUNCH(false, 64); }
#undef TN_LAUNCH
  return lbmk_check_launch("gemm_tn");
}
