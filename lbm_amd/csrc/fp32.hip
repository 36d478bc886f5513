// fp32 mode: every activation, weight and intermediate in f32 (get_model(torch_dtype=torch.float32), the reference's
// CPU-runnable configuration).  The same operators as the bf16 path — implicit-GEMM convolution / linear layers,
// GroupNorm(+SiLU), LayerNorm, row softmax, GEGLU, the layout kernels — written for exactness first:
// products on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain), statistics in f64,
// library expf / sinf / cosf / erff instead of the hardware approximations.  Attention runs unfused on these
// pieces (scores GEMM -> row softmax -> PV GEMM).  Peak of the f32 matrix path is 157 TFLOP/s, 1/16 of bf16.
#include "common.h"
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDS_LD = 132;   // 132: rows 4 apart in k land 16 banks apart

// ---- implicit GEMM, f32 ---------------------------------------------------------------------------------
// out[m][n] = act(alpha * sum_k A(m,k) w[n][k] + bias + rowvec[b(m)][n] + residual[m][n]), A gathered from NHWC.
// 256 threads = 4 waves, each a 64x64 corner of the 128x128 tile as 2x2 MFMA tiles of 32x32; 16-deep K tiles staged
// through LDS k-major (the MFMA operand of lane l is element [k = l >> 5][row = l & 31]: one conflict-free
// ds_read_b32 per operand), next tile's global loads issued before the current tile's products.  (Two stage buffers with one
// barrier per K tile measured the same: large layers run at 100-110 TFLOP/s, which is the matrix pipe at the clock the chip
// holds under this load.)
// K is summed in chunks of chunk_tiles(K) K tiles: products of a chunk run as one fmaf chain in the accumulators, and the
// chunk sums are added in chunk order.  SPLIT: workgroup z of grid.z owns chunk z and writes its raw chunk sum to
// d.f32_part[z][M][N]; splitk_finish_f32_kernel adds the chunks in the same order and applies the epilogue, so both forms
// return the same bits.  The chunk depth is a function of the LAYER -- K, N and the pixels per image (1 for a linear
// layer) -- never of the row count: 128 deep, doubled until there are at most clamp(1024 / output tiles of one image, 1, 32)
// chunks.  A 256-token, 1280-wide, 1280-deep layer is 2 x 10 tiles x 10 chunks = 200 workgroups for 256 CUs; a 64x64
// convolution (96 tiles per image) gets 6 chunks instead of 23 (a quarter of the partial-sum traffic; a CU holds four
// workgroups, so 1024 fill the chip); the VAE's
// megapixel convolutions get one (no second accumulator set: three waves per SIMD instead of two).
__host__ __device__ inline int chunk_tiles(int K, int N, int ohw) {
  const int nkt = (K + BK - 1) / BK;
  const long tiles_img = (long)((ohw + BM - 1) / BM) * ((N + BN - 1) / BN);
  long most = 1024 / tiles_img;
  most = most > 32 ? 32 : most < 1 ? 1 : most;
  int ct = 8;
  while ((nkt + ct - 1) / ct > most) ct *= 2;
  return ct;
}

// alpha, bias, per-image row vector, residual, activation of one output element (one definition for both forms)
__device__ __forceinline__ float f32_epilogue(const lbmk_igemm_desc& d, float sum, int m, int n, int ohw, const float* res) {
  float x = fmaf(d.alpha, sum, (d.bias && !d.bias_per_m) ? d.bias[n] : 0.f);
  if (d.bias && d.bias_per_m) x += d.bias[m];
  if (d.rowvec) x += ((const float*)d.rowvec)[(int64_t)(m / ohw) * d.rowvec_ld + n];
  if (res) x += res[(int64_t)m * d.ldr + n];
  if (d.act == LBMK_ACT_SILU) x = x / (1.0f + expf(-x));
  else if (d.act == LBMK_ACT_RELU) x = fmaxf(x, 0.f);
  return x;
}

// CHUNKED (unsplit form, more than one chunk): a second accumulator set holds the sum of the finished chunks; 64 more registers, two
// waves per SIMD instead of three (-3 % on the VAE's convolutions; capping the registers at three waves spills 31 of them
// into the loop, -8 %).
template <bool SPLIT, bool CHUNKED>
__global__ __launch_bounds__(256, CHUNKED ? 2 : 3) void igemm_f32_kernel(const lbmk_igemm_desc d) {
  __shared__ float As[BK][LDS_LD];
  __shared__ float Ws[BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* a = (const float*)d.a;
  const float* w = (const float*)d.w;
  float* out = (float*)d.out;
  const float* res = (const float*)d.residual;
  if constexpr (!SPLIT) {
    const int z = blockIdx.z;
    if (d.batch_inner > 1) {
      const int64_t zo = z / d.batch_inner, zi = z % d.batch_inner;
      a += zo * d.sa + zi * d.sa2;
      w += zo * d.sw + zi * d.sw2;
      out += zo * d.so + zi * d.so2;
    } else {
      a += (int64_t)z * d.sa;
      w += (int64_t)z * d.sw;
      out += (int64_t)z * d.so;
      if (res) res += (int64_t)z * d.sr;
    }
  }
  // staging: thread -> (row r of the tile, r + 64) x (4 consecutive k at g * 4)
  const int r = tid >> 2, g = tid & 3;
  const int ohw = d.OH * d.OW;
  int64_t img_base[2];
  int iy0[2], ix0[2];
  bool mrow[2];
  const float* wrow[2];
  bool nrow[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int m = m0 + r + 64 * p;
    mrow[p] = m < d.M;
    const int mm = mrow[p] ? m : 0;
    const int b = mm / ohw, pix = mm - b * ohw;
    const int oy = pix / d.OW, ox = pix - oy * d.OW;
    img_base[p] = (int64_t)b * d.IH * d.IW;
    iy0[p] = oy * d.stride - d.pad_t;
    ix0[p] = ox * d.stride - d.pad_l;
    const int n = n0 + r + 64 * p;
    nrow[p] = n < d.N;
    wrow[p] = w + (int64_t)(nrow[p] ? n : 0) * d.Kp;
  }
  const int IHe = d.up2 ? 2 * d.IH : d.IH, IWe = d.up2 ? 2 * d.IW : d.IW;
  f32x4 ra[2], rw[2];
  auto fetch = [&](int kt) {
    const int k0 = kt * BK + g * 4;
    const bool kin = k0 < d.K;
    const int tap = kin ? k0 / d.Cin : 0;
    const int c = k0 - tap * d.Cin;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int iy = iy0[p] + kh, ix = ix0[p] + kw;
      const bool ok = kin && mrow[p] && iy >= 0 && iy < IHe && ix >= 0 && ix < IWe;
      if (d.up2) { iy >>= 1; ix >>= 1; }
      ra[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) ra[p] = *(const f32x4*)(a + (img_base[p] + (int64_t)iy * d.IW + ix) * d.a_cs + d.a_coff + c);
      rw[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kin && nrow[p]) rw[p] = *(const f32x4*)(wrow[p] + k0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[g * 4 + e][r + 64 * p] = ra[p][e];
        Ws[g * 4 + e][r + 64 * p] = rw[p][e];
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = lane & 31, lk = lane >> 5;
  const int nkt_all = (d.K + BK - 1) / BK;
  const int CHUNK_TILES = chunk_tiles(d.K, d.N, d.OH * d.OW);
  const int kt_begin = SPLIT ? (int)blockIdx.z * CHUNK_TILES : 0;
  const int kt_end = SPLIT ? (kt_begin + CHUNK_TILES < nkt_all ? kt_begin + CHUNK_TILES : nkt_all) : nkt_all;
  // `first`: the first K tile of a chunk starts its accumulators from the constant zero (the matrix instruction takes an
  // inline 0 as its addend), so nothing is cleared between chunks
  auto tile_products = [&](auto first) {
    f32x16 zero;
#pragma unroll
    for (int v = 0; v < 16; ++v) zero[v] = 0.f;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = kk * 2 + lk;
      const float a0 = As[k][wm + lr], a1 = As[k][wm + 32 + lr];
      const float b0 = Ws[k][wn + lr], b1 = Ws[k][wn + 32 + lr];
      const bool z = decltype(first)::value && kk == 0;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, z ? zero : acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, z ? zero : acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, z ? zero : acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, z ? zero : acc[1][1], 0, 0, 0);
    }
  };
  const std::true_type FIRST{};
  const std::false_type REST{};
  fetch(kt_begin);
  if constexpr (SPLIT) {
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      __syncthreads();            // every wave is done reading the previous tile
      stage();
      __syncthreads();
      if (kt + 1 < kt_end) fetch(kt + 1);
      if (kt == kt_begin) tile_products(FIRST);
      else tile_products(REST);
    }
    // raw chunk sum: lane = output column, registers = rows (row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5))
    float* part = d.f32_part + (int64_t)blockIdx.z * d.M * d.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        if (n >= d.N) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lk;
          if (m < d.M) part[(int64_t)m * d.N + n] = acc[i][j][v];
        }
      }
  } else {
    // chunk sums are added in chunk order (same association as the split form's finishing pass)
    f32x16 tot[2][2];
    constexpr bool chunked = CHUNKED;   // one chunk (host's choice): the accumulators are the result
    if (chunked) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) tot[i][j][v] = 0.f;
    }
    for (int c0 = 0; c0 < nkt_all; c0 += CHUNK_TILES) {
      const int c1 = c0 + CHUNK_TILES < nkt_all ? c0 + CHUNK_TILES : nkt_all;
      for (int kt = c0; kt < c1; ++kt) {
        __syncthreads();
        stage();
        __syncthreads();
        if (kt + 1 < nkt_all) fetch(kt + 1);
        if (kt == c0) tile_products(FIRST);
        else tile_products(REST);
      }
      if (chunked) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) tot[i][j][v] += acc[i][j][v];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + j * 32 + lr;
        if (n >= d.N) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lk;
          if (m >= d.M) continue;
          out[(int64_t)m * d.ldo + n] = f32_epilogue(d, chunked ? tot[i][j][v] : acc[i][j][v], m, n, ohw, res);
        }
      }
  }
}

// sum of the chunk partials in chunk order + epilogue; VEC consecutive output columns per thread (VEC = 4: N % 4 == 0, so
// every 16-byte partial load is aligned), the chunk loop unrolled four deep so that four loads are in flight per thread
template <int VEC>
__global__ __launch_bounds__(256) void splitk_finish_f32_kernel(const lbmk_igemm_desc d) {
  const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  const int64_t MN = (int64_t)d.M * d.N;
  if (idx >= MN) return;
  const int m = (int)(idx / d.N), n = (int)(idx - (int64_t)m * d.N);
  typedef __attribute__((ext_vector_type(VEC))) float vec_t;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  const float* p = d.f32_part + idx;
  int z = 0;
  for (; z + 4 <= d.f32_chunks; z += 4) {
    vec_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const vec_t*)(p + (z + u) * MN);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += v[u][e];
  }
  for (; z < d.f32_chunks; ++z) {
    const vec_t v = *(const vec_t*)(p + z * MN);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] += v[e];
  }
  const int ohw = d.OH * d.OW;
#pragma unroll
  for (int e = 0; e < VEC; ++e)
    ((float*)d.out)[(int64_t)m * d.ldo + n + e] = f32_epilogue(d, s[e], m, n + e, ohw, (const float*)d.residual);
}

// ---- GroupNorm(+SiLU), f32: slab partial sums in f64 -> mean / rstd -> apply ------------------------------
constexpr int GN_SLABS = 64;

__global__ __launch_bounds__(256) void gn32_stats_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                         int HW, int C, int G) {
  __shared__ double red[2][4];
  const int slab = blockIdx.x, bg = blockIdx.y;
  const int b = bg / G, g = bg - b * G, cpg = C / G;
  const int rows = (HW + GN_SLABS - 1) / GN_SLABS;
  const int r0 = slab * rows, r1 = min(r0 + rows, HW);
  const float* base = x + (int64_t)b * HW * C + g * cpg;
  double s = 0.0, ss = 0.0;
  if ((cpg & 3) == 0 && (C & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
    // 16 bytes per load: a group's channels of one pixel are cpg / 4 consecutive float4
    const int c4n = cpg >> 2;
    const int64_t n4 = (int64_t)max(r1 - r0, 0) * c4n;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const int64_t row = r0 + i / c4n;
      const int c = (int)(i % c4n) * 4;
      const f32x4 v4 = *(const f32x4*)(base + row * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double v = (double)v4[e];
        s += v;
        ss += v * v;
      }
    }
  } else {
    const int64_t n = (int64_t)max(r1 - r0, 0) * cpg;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const int64_t row = r0 + i / cpg;
      const int c = (int)(i % cpg);
      const double v = (double)base[row * C + c];
      s += v;
      ss += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((int64_t)bg * GN_SLABS + slab) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[((int64_t)bg * GN_SLABS + slab) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void gn32_finalize_kernel(const double* __restrict__ partial, float* __restrict__ stats, int BG,
                                     double inv_count, double eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BG) return;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < GN_SLABS; ++k) { s += partial[((int64_t)i * GN_SLABS + k) * 2]; ss += partial[((int64_t)i * GN_SLABS + k) * 2 + 1]; }
  const double mean = s * inv_count;
  double var = ss * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + eps));
}

__global__ void gn32_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, int64_t total4,
                                  int HW, int C, int G, int silu) {
  const int cpg = C / G, c4n = C / 4;
  const bool vec_affine = (cpg & 3) == 0 && (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / c4n;
    const int c0 = (int)(i - pix * c4n) * 4;
    const int b = (int)(pix / HW);
    const f32x4 v = *(const f32x4*)(x + i * 4);
    f32x4 o;
    if (vec_affine) {   // the four channels share a group: one statistics pair, 16-byte scale / shift loads
      const int g = c0 / cpg;
      const float mean = stats[((int64_t)b * G + g) * 2], rstd = stats[((int64_t)b * G + g) * 2 + 1];
      const f32x4 g4 = gamma ? *(const f32x4*)(gamma + c0) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 b4 = beta ? *(const f32x4*)(beta + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f = (v[e] - mean) * rstd;
        f = f * g4[e] + b4[e];
        if (silu) f = f / (1.0f + expf(-f));
        o[e] = f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e, g = c / cpg;
        const float mean = stats[((int64_t)b * G + g) * 2], rstd = stats[((int64_t)b * G + g) * 2 + 1];
        float f = (v[e] - mean) * rstd;
        f = f * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        if (silu) f = f / (1.0f + expf(-f));
        o[e] = f;
      }
    }
    *(f32x4*)(y + i * 4) = o;
  }
}

// ---- LayerNorm, f32: one wave per row, the row in registers (C <= 2048) ------------------------------------
__global__ __launch_bounds__(256) void ln32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * C;
  float v[32];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = lane + 64 * i;
    if (c < C) { const float dlt = v[i] - mean; q += dlt * dlt; }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  float* yr = y + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = lane + 64 * i;
    if (c < C) yr[c] = (v[i] - mean) * rstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  }
}

// same, 16-byte accesses (C % 4 == 0, 16-byte aligned rows): a quarter of the memory instructions per row
__global__ __launch_bounds__(256) void ln32_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4* xr = (const f32x4*)(x + (int64_t)row * C);
  const int c4n = C >> 2;
  f32x4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < c4n ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (lane + 64 * i < c4n) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dlt = v[i][e] - mean; q += dlt * dlt; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  f32x4* yr = (f32x4*)(y + (int64_t)row * C);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    if (c < c4n) {
      const f32x4 g4 = gamma ? ((const f32x4*)gamma)[c] : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 b4 = beta ? ((const f32x4*)beta)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
      yr[c] = o;
    }
  }
}

// ---- row softmax f32 -> f32: one workgroup per row, three passes over the row --------------------------------
__global__ __launch_bounds__(256) void softmax32_kernel(const float* __restrict__ s, float* __restrict__ p, int cols,
                                                        int64_t ld_in, int64_t ld_out) {
  __shared__ float red[4];
  const float* sr = s + (int64_t)blockIdx.x * ld_in;
  float* pr = p + (int64_t)blockIdx.x * ld_out;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) sum += expf(sr[c] - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < cols; c += 256) pr[c] = expf(sr[c] - mx) * inv;
}

// ---- GEGLU f32: h [rows][2F] -> y [rows][F] = h[:, :F] * gelu_erf(h[:, F:]) ---------------------------------
__global__ void geglu32_kernel(const float* __restrict__ h, float* __restrict__ y, int64_t total, int F) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / F;
    const int c = (int)(i - row * F);
    const float val = h[row * 2 * F + c], gate = h[row * 2 * F + F + c];
    y[i] = val * (0.5f * gate * (1.0f + erff(gate * 0.70710678118654752440f)));
  }
}

// ---- layout / glue, f32 activations --------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ldx(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldx<bf16_t>(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ldx<float>(const float* p, int64_t i) { return p[i]; }
template <typename T> __device__ __forceinline__ void stx(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stx<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stx<float>(float* p, int64_t i, float v) { p[i] = v; }

template <typename T>
__global__ void nchw_to_nhwc32_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int64_t HW, int Cp) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const T* xb = x + (int64_t)b * C * HW + p;
  float* yp = y + ((int64_t)b * HW + p) * Cp;
  for (int c = 0; c < Cp; ++c) yp[c] = c < C ? ldx(xb, (int64_t)c * HW) : 0.f;
}

template <typename T>
__global__ void nhwc_to_nchw32_kernel(const float* __restrict__ x, T* __restrict__ y, int C, int64_t HW, int Cs) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const float* xp = x + ((int64_t)b * HW + p) * Cs;
  T* yb = y + (int64_t)b * C * HW + p;
  for (int c = 0; c < C; ++c) stx(yb, (int64_t)c * HW, xp[c]);
}

__global__ void copy_channels32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t total, int nch,
                                       int src_cs, int src_off, int dst_cs, int dst_off) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / nch;
    const int c = (int)(i - p * nch) * 4;
    *(f32x4*)(dst + p * dst_cs + dst_off + c) = *(const f32x4*)(src + p * src_cs + src_off + c);
  }
}

__global__ void posterior_sample32_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                          float* __restrict__ z, int Cz, int64_t HW, int ms, float shift, float scale) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const float* mp = mom + ((int64_t)b * HW + p) * ms;
  for (int c = 0; c < Cz; ++c) {
    const float logvar = fminf(fmaxf(mp[Cz + c], -30.f), 20.f);
    const int64_t o = ((int64_t)b * Cz + c) * HW + p;
    z[o] = ((mp[c] + expf(0.5f * logvar) * eps[o]) - shift) * scale;
  }
}

__global__ void timestep_embedding32_kernel(const float* __restrict__ t, float* __restrict__ out, int dim) {
  const int b = blockIdx.x, half = dim >> 1;
  const float tv = t[b];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float arg = tv * expf(-9.210340371976184f * (float)i / (float)half);
    out[(int64_t)b * dim + i] = cosf(arg);
    out[(int64_t)b * dim + half + i] = sinf(arg);
  }
}

__global__ void unary32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int mode, float a, float bsh) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = x[i];
    y[i] = mode == 0 ? f / (1.0f + expf(-f)) : f * a + bsh;
  }
}

__global__ void tile_accumulate32_kernel(const float* __restrict__ tile, float* __restrict__ acc, float* __restrict__ wsum,
                                         const float* __restrict__ wy, const float* __restrict__ wx, int C, int Cs,
                                         int th_full, int tw_full, int th, int tw, int H, int W, int y0, int x0) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= tw || y >= th) return;
  const int oy = y0 + y, ox = x0 + x;
  if (oy >= H || ox >= W) return;
  const float wgt = wy[y] * wx[x];
  const float* tp = tile + (((int64_t)b * th_full + y) * tw_full + x) * Cs;
  for (int c = 0; c < C; ++c) acc[(((int64_t)b * C + c) * H + oy) * W + ox] += tp[c] * wgt;
  if (b == 0) wsum[(int64_t)oy * W + ox] += wgt;
}

inline unsigned grid1d(int64_t n, int block, int64_t cap = 8192) {
  int64_t gsz = (n + block - 1) / block;
  if (gsz > cap) gsz = cap;
  if (gsz < 1) gsz = 1;
  return (unsigned)gsz;
}

}  // namespace

// called by lbmk_igemm when the descriptor's in_f32 is set
int lbmk_igemm_f32_launch(const lbmk_igemm_desc* d, void* stream) {
  LBMK_REQUIRE(d->a && d->w && d->out, "igemm (f32): null pointer");
  LBMK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->Kp >= d->K, "igemm (f32): M=%d N=%d K=%d Kp=%d", d->M, d->N, d->K, d->Kp);
  LBMK_REQUIRE(d->Cin > 0 && d->Cin % 4 == 0 && d->K == d->KH * d->KW * d->Cin && d->Kp % 4 == 0 && d->a_cs % 4 == 0 &&
                   d->a_coff % 4 == 0,
               "igemm (f32): Cin=%d (multiple of 4), K=%d = KH*KW*Cin, Kp=%d, a_cs=%d, a_coff=%d multiples of 4", d->Cin, d->K,
               d->Kp, d->a_cs, d->a_coff);
  LBMK_REQUIRE(lbmk_aligned(d->a, 16) && lbmk_aligned(d->w, 16), "igemm (f32): a / w must be 16-byte aligned");
  LBMK_REQUIRE(d->out_f32 == 1, "igemm (f32): the output is f32 (out_f32 = 1)");
  LBMK_REQUIRE(d->act == LBMK_ACT_NONE || d->act == LBMK_ACT_SILU || d->act == LBMK_ACT_RELU,
               "igemm (f32): GEGLU is not fused on this path (lbmk_geglu_f32 after the projection)");
  LBMK_REQUIRE(!d->gn_partial && d->k_order == 0, "igemm (f32): no fused GroupNorm statistics, k_order 0 only");
  LBMK_REQUIRE(d->OH > 0 && d->OW > 0 && d->IH > 0 && d->IW > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0, "igemm (f32): geometry");
  LBMK_REQUIRE(d->M % (d->OH * d->OW) == 0, "igemm (f32): M=%d is not a whole number of %dx%d images", d->M, d->OH, d->OW);
  LBMK_REQUIRE(d->batch >= 1 && d->batch <= 65535, "igemm (f32): batch=%d", d->batch);
  LBMK_REQUIRE(d->batch_inner <= 1 || (!d->residual && d->batch % d->batch_inner == 0), "igemm (f32): batch_inner");
  dim3 grid((d->M + BM - 1) / BM, (d->N + BN - 1) / BN, d->batch);
  LBMK_REQUIRE(grid.y <= 65535, "igemm (f32): N=%d too large for one launch", d->N);
  if (d->f32_chunks > 1) {
    LBMK_REQUIRE(d->batch == 1 && d->batch_inner <= 1 && d->f32_part && lbmk_aligned(d->f32_part, 4) &&
                     d->f32_chunks == lbmk_igemm_f32_chunks(d->K, d->N, d->OH * d->OW),
                 "igemm (f32): a split launch takes batch 1, a partial buffer and f32_chunks = %d (got %d)",
                 lbmk_igemm_f32_chunks(d->K, d->N, d->OH * d->OW), d->f32_chunks);
    grid.z = d->f32_chunks;
    hipLaunchKernelGGL((igemm_f32_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, *d);
    if (int rc = lbmk_check_launch("igemm (f32, chunks)")) return rc;
    const int64_t MN = (int64_t)d->M * d->N;
    if (d->N % 4 == 0 && lbmk_aligned(d->f32_part, 16))
      hipLaunchKernelGGL(splitk_finish_f32_kernel<4>, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    else
      hipLaunchKernelGGL(splitk_finish_f32_kernel<1>, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    return lbmk_check_launch("igemm (f32, finish)");
  }
  if (lbmk_igemm_f32_chunks(d->K, d->N, d->OH * d->OW) > 1) hipLaunchKernelGGL((igemm_f32_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL((igemm_f32_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, *d);
  return lbmk_check_launch("igemm (f32)");
}

extern "C" int lbmk_igemm_f32_chunks(int K, int N, int pixels_per_image) {
  const int ct = chunk_tiles(K, N, pixels_per_image), nkt = (K + BK - 1) / BK;
  return (nkt + ct - 1) / ct;
}

// few output tiles against many chunks: the chunked launch fills the chip, and its partial traffic (2 x chunks x M x N x 4
// bytes) is small beside the products
extern "C" int lbmk_igemm_f32_split_pays(int M, int N, int K, int pixels_per_image, int batch) {
  if (batch != 1 || M <= 0 || N <= 0) return 0;
  const int chunks = lbmk_igemm_f32_chunks(K, N, pixels_per_image);
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return chunks >= 2 && tiles <= lbmk_num_cus() ? 1 : 0;
}

extern "C" int64_t lbmk_groupnorm_f32_scratch_floats(int B, int G) { return (int64_t)B * G * (GN_SLABS * 4 + 2); }

extern "C" int lbmk_groupnorm_f32(const void* x, void* y, const float* gamma, const float* beta, float* scratch, int B,
                                  int HW, int C, int G, float eps, int silu, void* stream) {
  LBMK_REQUIRE(x && y && scratch && B > 0 && HW > 0 && G > 0 && C % G == 0 && C % 4 == 0, "groupnorm_f32: B=%d HW=%d C=%d G=%d", B, HW, C, G);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(y, 16) && lbmk_aligned(scratch, 8), "groupnorm_f32: alignment");
  LBMK_REQUIRE((int64_t)B * G <= 65535, "groupnorm_f32: B*G too large");
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)scratch;
  float* stats = scratch + (int64_t)B * G * GN_SLABS * 4;
  hipLaunchKernelGGL(gn32_stats_kernel, dim3(GN_SLABS, B * G), dim3(256), 0, s, (const float*)x, partial, HW, C, G);
  hipLaunchKernelGGL(gn32_finalize_kernel, dim3((B * G + 63) / 64), dim3(64), 0, s, partial, stats, B * G,
                     1.0 / ((double)HW * (double)(C / G)), (double)eps);
  const int64_t total4 = (int64_t)B * HW * C / 4;
  hipLaunchKernelGGL(gn32_apply_kernel, dim3(grid1d(total4, 256)), dim3(256), 0, s, (const float*)x, (float*)y, stats, gamma,
                     beta, total4, HW, C, G, silu);
  return lbmk_check_launch("groupnorm_f32");
}

extern "C" int lbmk_layernorm_f32(const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps,
                                  void* stream) {
  LBMK_REQUIRE(x && y && rows > 0 && C > 0 && C <= 2048, "layernorm_f32: rows=%d C=%d (<= 2048)", rows, C);
  if (C % 4 == 0 && lbmk_aligned(x, 16) && lbmk_aligned(y, 16) && (!gamma || lbmk_aligned(gamma, 16)) && (!beta || lbmk_aligned(beta, 16)))
    hipLaunchKernelGGL(ln32_vec_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, gamma,
                       beta, rows, C, eps);
  else
    hipLaunchKernelGGL(ln32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, gamma,
                       beta, rows, C, eps);
  return lbmk_check_launch("layernorm_f32");
}

extern "C" int lbmk_softmax_rows_f32(const float* s, float* p, int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                                     void* stream) {
  LBMK_REQUIRE(s && p && rows > 0 && rows < (1LL << 31) && cols > 0 && ld_in >= cols && ld_out >= cols, "softmax_rows_f32: sizes");
  hipLaunchKernelGGL(softmax32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, p, cols, ld_in, ld_out);
  return lbmk_check_launch("softmax_rows_f32");
}

extern "C" int lbmk_geglu_f32(const float* h, float* y, int64_t rows, int F, void* stream) {
  LBMK_REQUIRE(h && y && rows > 0 && F > 0, "geglu_f32: sizes");
  hipLaunchKernelGGL(geglu32_kernel, dim3(grid1d(rows * F, 256)), dim3(256), 0, (hipStream_t)stream, h, y, rows * F, F);
  return lbmk_check_launch("geglu_f32");
}

extern "C" int lbmk_nchw_to_nhwc_f32(const void* x, float* y, int B, int C, int H, int W, int Cp, int in_dtype, void* stream) {
  LBMK_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 4 == 0 && (in_dtype | 1) == 1,
               "nchw_to_nhwc_f32: sizes / dtype");
  const int64_t HW = (int64_t)H * W;
  dim3 grid((unsigned)((HW + 255) / 256), B);
  if (in_dtype == 0)
    hipLaunchKernelGGL(nchw_to_nhwc32_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, C, HW, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc32_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, y, C, HW, Cp);
  return lbmk_check_launch("nchw_to_nhwc_f32");
}

extern "C" int lbmk_nhwc_to_nchw_f32(const float* x, void* y, int B, int C, int H, int W, int Cs, int out_dtype, void* stream) {
  LBMK_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && Cs >= C && (out_dtype | 1) == 1,
               "nhwc_to_nchw_f32: sizes / dtype");
  const int64_t HW = (int64_t)H * W;
  dim3 grid((unsigned)((HW + 255) / 256), B);
  if (out_dtype == 0)
    hipLaunchKernelGGL(nhwc_to_nchw32_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, C, HW, Cs);
  else
    hipLaunchKernelGGL(nhwc_to_nchw32_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)y, C, HW, Cs);
  return lbmk_check_launch("nhwc_to_nchw_f32");
}

extern "C" int lbmk_copy_channels_f32(const float* src, float* dst, int64_t pixels, int C, int src_cs, int src_off, int dst_cs,
                                      int dst_off, void* stream) {
  LBMK_REQUIRE(src && dst && pixels > 0 && C > 0, "copy_channels_f32: sizes");
  LBMK_REQUIRE(C % 4 == 0 && src_cs % 4 == 0 && src_off % 4 == 0 && dst_cs % 4 == 0 && dst_off % 4 == 0 &&
                   src_off + C <= src_cs && dst_off + C <= dst_cs && lbmk_aligned(src, 16) && lbmk_aligned(dst, 16),
               "copy_channels_f32: channel counts / offsets must be multiples of 4 inside their strides");
  const int nch = C / 4;
  hipLaunchKernelGGL(copy_channels32_kernel, dim3(grid1d(pixels * nch, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     pixels * nch, nch, src_cs, src_off, dst_cs, dst_off);
  return lbmk_check_launch("copy_channels_f32");
}

extern "C" int lbmk_posterior_sample_f32(const float* moments, const float* eps, float* z, int B, int Cz, int HW, int ms,
                                         float shift, float scale, void* stream) {
  LBMK_REQUIRE(moments && eps && z && B > 0 && B <= 65535 && Cz > 0 && HW > 0 && ms >= 2 * Cz, "posterior_sample_f32: sizes");
  hipLaunchKernelGGL(posterior_sample32_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream,
                     moments, eps, z, Cz, (int64_t)HW, ms, shift, scale);
  return lbmk_check_launch("posterior_sample_f32");
}

extern "C" int lbmk_timestep_embedding_f32(const float* t, float* out, int B, int dim, void* stream) {
  LBMK_REQUIRE(t && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding_f32: sizes");
  hipLaunchKernelGGL(timestep_embedding32_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, t, out, dim);
  return lbmk_check_launch("timestep_embedding_f32");
}

extern "C" int lbmk_silu_f32(const float* x, float* y, int64_t n, void* stream) {
  LBMK_REQUIRE(x && y && n > 0, "silu_f32: sizes");
  hipLaunchKernelGGL(unary32_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, 0, 0.f, 0.f);
  return lbmk_check_launch("silu_f32");
}

extern "C" int lbmk_scale_shift_f32(const float* x, float* y, int64_t n, float a, float b, void* stream) {
  LBMK_REQUIRE(x && y && n > 0, "scale_shift_f32: sizes");
  hipLaunchKernelGGL(unary32_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, 1, a, b);
  return lbmk_check_launch("scale_shift_f32");
}

extern "C" int lbmk_tile_accumulate_f32(const float* tile, float* acc, float* wsum, const float* wy, const float* wx, int B,
                                        int C, int Cs, int th_full, int tw_full, int th, int tw, int H, int W, int y0, int x0,
                                        void* stream) {
  LBMK_REQUIRE(tile && acc && wsum && wy && wx && B > 0 && B <= 65535 && C > 0 && Cs >= C && th > 0 && tw > 0 && th <= th_full &&
                   tw <= tw_full && th <= 65535 && y0 >= 0 && x0 >= 0,
               "tile_accumulate_f32: sizes");
  hipLaunchKernelGGL(tile_accumulate32_kernel, dim3((tw + 255) / 256, th, B), dim3(256), 0, (hipStream_t)stream, tile, acc, wsum,
                     wy, wx, C, Cs, th_full, tw_full, th, tw, H, W, y0, x0);
  return lbmk_check_launch("tile_accumulate_f32");
}
