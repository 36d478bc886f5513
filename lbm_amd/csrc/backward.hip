// Data-movement and reduction kernels of the training (backward) path, all HBM-bound:
//   transpose      [R][C] -> [C][R]             operands of the reduction-over-rows weight-gradient GEMM
//   im2col_t       conv gather, transposed      wgrad of a convolution:  dW = dY^T . col(X)
//   col2im         conv scatter as a gather     dgrad of strided / upsampled convolutions
//   colsum         per-group column sums        bias and time-embedding gradients
// The GEMMs themselves are lbmk_igemm launches on these re-laid-out operands.
#include "common.h"

namespace {

// out[c][r] = in[r][c]; 64x64 tiles through LDS, 16-byte accesses on both sides.
// LDS tile is [64 rows][64 + 8 cols] bf16: the 16-byte pad staggers the column reads across banks.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                        int64_t R, int C, int64_t ld_in, int64_t ld_out,
                                                        int64_t s_in, int64_t s_out) {
  __shared__ bf16_t tile[64][72];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  in += (int64_t)blockIdx.z * s_in;
  out += (int64_t)blockIdx.z * s_out;
  // load: 8 threads cover one 64-column row (16 B each); 32 rows per pass
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = (t >> 3) + 32 * pass, cl = (t & 7) * 8;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v.v[e] = 0;
    if (r0 + rl < R && c0 + cl < C) v = *(const u16x8*)(in + (r0 + rl) * ld_in + c0 + cl);
    *(u16x8*)&tile[rl][cl] = v;
  }
  __syncthreads();
  // store: 8 threads cover 64 rows of one output row (= input column)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (t >> 3) + 32 * pass, rl = (t & 7) * 8;
    if (c0 + cl < C && r0 + rl < R) {
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = tile[rl + e][cl];
      *(u16x8*)(out + (int64_t)(c0 + cl) * ld_out + r0 + rl) = v;
    }
  }
}

struct ConvGeom {
  int B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, up2, Cin, a_cs, a_coff;
};

// out[(tap*Cin + c)][p] = X[src(p, tap)][c]  (0 where the tap falls outside the image), p = output pixel.
// grid: (pixel tiles of 64, channel tiles of 64, taps)
__global__ __launch_bounds__(256) void im2col_t_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                       ConvGeom g, int64_t M, int64_t ld_out) {
  __shared__ bf16_t tile[64][72];
  const int t = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tap = blockIdx.z;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int ohw = g.OH * g.OW;
  const int IHe = g.up2 ? 2 * g.IH : g.IH, IWe = g.up2 ? 2 * g.IW : g.IW;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int pl = (t >> 3) + 32 * pass, cl = (t & 7) * 8;
    const int64_t p = p0 + pl;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v.v[e] = 0;
    if (p < M && c0 + cl < g.Cin) {
      const int b = (int)(p / ohw), rem = (int)(p - (int64_t)b * ohw);
      const int oy = rem / g.OW, ox = rem - oy * g.OW;
      int iy = oy * g.stride - g.pad_t + kh, ix = ox * g.stride - g.pad_l + kw;
      if ((unsigned)iy < (unsigned)IHe && (unsigned)ix < (unsigned)IWe) {
        if (g.up2) { iy >>= 1; ix >>= 1; }
        v = *(const u16x8*)(x + ((int64_t)b * g.IH * g.IW + (int64_t)iy * g.IW + ix) * g.a_cs + g.a_coff + c0 + cl);
      }
    }
    *(u16x8*)&tile[pl][cl] = v;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (t >> 3) + 32 * pass, pl = (t & 7) * 8;
    if (c0 + cl < g.Cin && p0 + pl < M) {
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = tile[pl + e][cl];
      *(u16x8*)(out + ((int64_t)tap * g.Cin + c0 + cl) * ld_out + p0 + pl) = v;
    }
  }
}

// dX[q][c] = sum over taps and over the output pixels p whose tap reads input pixel q of dcol[p][tap*Cin + c].
// One thread per (input pixel, 8-channel chunk); written as a gather, so no atomics.
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx,
                                                     ConvGeom g, int64_t npix_in, int64_t ld_col, int dx_cs) {
  const int chunks = g.Cin >> 3;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= npix_in * chunks) return;
  const int ch = (int)(gid % chunks) * 8;
  const int64_t q = gid / chunks;
  const int ihw = g.IH * g.IW;
  const int b = (int)(q / ihw), rem = (int)(q - (int64_t)b * ihw);
  const int iy = rem / g.IW, ix = rem - iy * g.IW;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int nu = g.up2 ? 2 : 1;
  for (int kh = 0; kh < g.KH; ++kh)
    for (int uy = 0; uy < nu; ++uy) {
      const int ty = (g.up2 ? 2 * iy + uy : iy) + g.pad_t - kh;
      if (ty < 0 || ty % g.stride) continue;
      const int oy = ty / g.stride;
      if (oy >= g.OH) continue;
      for (int kw = 0; kw < g.KW; ++kw)
        for (int ux = 0; ux < nu; ++ux) {
          const int tx = (g.up2 ? 2 * ix + ux : ix) + g.pad_l - kw;
          if (tx < 0 || tx % g.stride) continue;
          const int ox = tx / g.stride;
          if (ox >= g.OW) continue;
          const int64_t p = ((int64_t)b * g.OH + oy) * g.OW + ox;
          const u16x8 v = *(const u16x8*)(dcol + p * ld_col + (kh * g.KW + kw) * g.Cin + ch);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += bf2f(v.v[e]);
        }
    }
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.v[e] = f2bf(acc[e]);
  *(u16x8*)(dx + q * dx_cs + ch) = o;
}

// partial[s][g][n] = sum of x[m][n] over the rows m of segment s of group g (groups = rows_per_group consecutive rows)
// grid: (column chunks of 64, groups, segments); 8 lanes cover the 64 columns, 32 row lanes stride the segment.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                            int64_t rows_per_group, int N, int64_t ld, int nseg) {
  __shared__ float red[32][65];
  const int t = threadIdx.x;
  const int cl = (t & 7) * 8, rl = t >> 3;
  const int c0 = blockIdx.x * 64, grp = blockIdx.y, seg = blockIdx.z;
  const int64_t seg_rows = (rows_per_group + nseg - 1) / nseg;
  const int64_t r0 = (int64_t)grp * rows_per_group + seg * seg_rows;
  const int64_t r1 = min(r0 + seg_rows, (int64_t)(grp + 1) * rows_per_group);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (c0 + cl < N) {
    for (int64_t r = r0 + rl; r < r1; r += 32) {
      const u16x8 v = *(const u16x8*)(x + r * ld + c0 + cl);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += bf2f(v.v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl + e] = s[e];
  __syncthreads();
  if (t < 64 && c0 + t < N) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) a += red[r][t];
    partial[((int64_t)seg * gridDim.y + grp) * N + c0 + t] = a;
  }
}

// out[i] = sum over segments of partial[s][i]; out dtype 0 bf16 / 1 f32.  32 columns per workgroup, 8 segment
// lanes per column with independent loads in flight; the 8 lane sums are added in a fixed order.
// out_b != null: elements from `split` on go to out_b[i - split] (two destinations of one reduction: dgamma | dbeta).
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, void* __restrict__ out,
                                                           int64_t GN, int nseg, int out_dtype, void* __restrict__ out_b = nullptr,
                                                           int64_t split = 0) {
  __shared__ float red[8][33];
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < GN) {
    int s = sl;
    for (; s + 24 < nseg; s += 32) {
      a0 += partial[(int64_t)s * GN + i];
      a1 += partial[(int64_t)(s + 8) * GN + i];
      a2 += partial[(int64_t)(s + 16) * GN + i];
      a3 += partial[(int64_t)(s + 24) * GN + i];
    }
    for (; s < nseg; s += 8) a0 += partial[(int64_t)s * GN + i];
  }
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (t < 32 && i < GN) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][t];
    void* dst = out;
    int64_t j = i;
    if (out_b && i >= split) { dst = out_b; j = i - split; }
    if (out_dtype) ((float*)dst)[j] = a;
    else ((bf16_t*)dst)[j] = f2bf(a);
  }
}

}  // namespace

extern "C" int lbmk_transpose(const void* in, void* out, int64_t R, int C, int64_t ld_in, int64_t ld_out,
                              int batch, int64_t s_in, int64_t s_out, void* stream) {
  LBMK_REQUIRE(in && out, "transpose: null pointer");
  LBMK_REQUIRE(R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0, "transpose: R=%lld C=%d must be multiples of 8", (long long)R, C);
  LBMK_REQUIRE(ld_in >= C && ld_out >= R && ld_in % 8 == 0 && ld_out % 8 == 0, "transpose: leading dims");
  LBMK_REQUIRE(lbmk_aligned(in, 16) && lbmk_aligned(out, 16), "transpose: misaligned");
  LBMK_REQUIRE((R + 63) / 64 < (1LL << 31) && (C + 63) / 64 <= 65535, "transpose: grid");
  LBMK_REQUIRE(batch >= 1 && batch <= 65535 && s_in % 8 == 0 && s_out % 8 == 0, "transpose: batch %d / strides", batch);
  dim3 grid((unsigned)((R + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, R, C,
                     ld_in, ld_out, s_in, s_out);
  return lbmk_check_launch("transpose");
}

static bool geom_ok(int B, int IH, int IW, int OH, int OW, int KH, int KW, int stride, int Cin, int a_cs, int a_coff) {
  return B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && ((KH == 1 && KW == 1) || (KH == 3 && KW == 3)) && stride >= 1 &&
         Cin > 0 && Cin % 8 == 0 && a_cs % 8 == 0 && a_coff % 8 == 0 && a_coff >= 0 && a_coff + Cin <= a_cs;
}

extern "C" int lbmk_im2col_t(const void* x, void* out, int B, int IH, int IW, int OH, int OW, int KH, int KW,
                             int stride, int pad_t, int pad_l, int up2, int Cin, int a_cs, int a_coff,
                             int64_t ld_out, void* stream) {
  LBMK_REQUIRE(x && out, "im2col_t: null pointer");
  LBMK_REQUIRE(geom_ok(B, IH, IW, OH, OW, KH, KW, stride, Cin, a_cs, a_coff), "im2col_t: geometry");
  const int64_t M = (int64_t)B * OH * OW;
  LBMK_REQUIRE(M % 8 == 0 && ld_out >= M && ld_out % 8 == 0, "im2col_t: M=%lld ld_out=%lld", (long long)M, (long long)ld_out);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(out, 16), "im2col_t: misaligned");
  ConvGeom g{B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, up2, Cin, a_cs, a_coff};
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((Cin + 63) / 64), (unsigned)(KH * KW));
  hipLaunchKernelGGL(im2col_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, g, M, ld_out);
  return lbmk_check_launch("im2col_t");
}

extern "C" int lbmk_col2im(const void* dcol, void* dx, int B, int IH, int IW, int OH, int OW, int KH, int KW,
                           int stride, int pad_t, int pad_l, int up2, int Cin, int64_t ld_col, int dx_cs,
                           void* stream) {
  LBMK_REQUIRE(dcol && dx, "col2im: null pointer");
  LBMK_REQUIRE(geom_ok(B, IH, IW, OH, OW, KH, KW, stride, Cin, dx_cs, 0), "col2im: geometry");
  LBMK_REQUIRE(ld_col >= (int64_t)KH * KW * Cin && ld_col % 8 == 0, "col2im: ld_col");
  LBMK_REQUIRE(lbmk_aligned(dcol, 16) && lbmk_aligned(dx, 16), "col2im: misaligned");
  ConvGeom g{B, IH, IW, OH, OW, KH, KW, stride, pad_t, pad_l, up2, Cin, dx_cs, 0};
  const int64_t npix = (int64_t)B * IH * IW;
  const int64_t threads = npix * (Cin / 8);
  LBMK_REQUIRE((threads + 255) / 256 < (1LL << 31), "col2im: grid");
  hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dcol, (bf16_t*)dx, g, npix, ld_col, dx_cs);
  return lbmk_check_launch("col2im");
}

namespace {
// few segments, many elements (split-K partials of a weight gradient): one thread per 4 consecutive elements,
// segments added in order
__global__ __launch_bounds__(256) void sum_partials_vec_kernel(const float* __restrict__ partial, void* __restrict__ out,
                                                               int64_t n4, int64_t n, int nseg, int out_dtype) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 a = *(const f32x4*)(partial + i * 4);
  for (int s = 1; s < nseg; ++s) a += *(const f32x4*)(partial + (int64_t)s * n + i * 4);
  if (out_dtype) *(f32x4*)((float*)out + i * 4) = a;
  else {
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = f2bf(a[e]);
    *(u16x4*)((bf16_t*)out + i * 4) = o;
  }
}
// split-K forward / input-gradient GEMM: out[m][n] = bf16(sum_s part[s][m][n] + bias[n] + residual[m][n]), 8 columns per thread
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int nseg, int64_t seg_stride,
                                                            const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            int ldr, bf16_t* __restrict__ out, int ldo, int M, int n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * n8) return;
  const int m = (int)(i / n8), c = (int)(i - (int64_t)m * n8) * 8;
  const int N = n8 * 8;
  const float* p0 = part + (int64_t)m * N + c;
  f32x4 a0 = *(const f32x4*)p0, a1 = *(const f32x4*)(p0 + 4);
  for (int s = 1; s < nseg; ++s) {
    a0 += *(const f32x4*)(p0 + s * seg_stride);
    a1 += *(const f32x4*)(p0 + s * seg_stride + 4);
  }
  if (bias) { a0 += *(const f32x4*)(bias + c); a1 += *(const f32x4*)(bias + c + 4); }
  u16x8 o;
  if (res) {
    const u16x8 r = *(const u16x8*)(res + (int64_t)m * ldr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a0[e] += bf2f(r.v[e]); a1[e] += bf2f(r.v[4 + e]); }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { o.v[e] = f2bf(a0[e]); o.v[4 + e] = f2bf(a1[e]); }
  *(u16x8*)(out + (int64_t)m * ldo + c) = o;
}
}  // namespace

extern "C" int lbmk_splitk_finish(const float* part, int nseg, int64_t seg_stride, const float* bias, const void* residual,
                                  int ldr, void* out, int ldo, int M, int N, void* stream) {
  LBMK_REQUIRE(part && out && nseg >= 1 && M > 0 && N > 0 && N % 8 == 0, "splitk_finish: nseg=%d M=%d N=%d", nseg, M, N);
  LBMK_REQUIRE(seg_stride >= (int64_t)M * N && seg_stride % 4 == 0 && ldo >= N && ldo % 8 == 0 && (!residual || (ldr >= N && ldr % 8 == 0)),
               "splitk_finish: strides");
  LBMK_REQUIRE(lbmk_aligned(part, 16) && lbmk_aligned(out, 16) && (!residual || lbmk_aligned(residual, 16)) && (!bias || lbmk_aligned(bias, 16)),
               "splitk_finish: 16-byte alignment");
  const int64_t n = (int64_t)M * (N / 8);
  LBMK_REQUIRE((n + 255) / 256 < (1LL << 31), "splitk_finish: grid");
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, nseg, seg_stride,
                     bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, M, N / 8);
  return lbmk_check_launch("splitk_finish");
}

extern "C" int lbmk_sum_partials(const float* partial, void* out, int64_t n, int nseg, int out_dtype, void* stream) {
  LBMK_REQUIRE(partial && out && n > 0 && nseg > 0, "sum_partials: n=%lld nseg=%d", (long long)n, nseg);
  LBMK_REQUIRE(out_dtype == 0 || out_dtype == 1, "sum_partials: out_dtype %d", out_dtype);
  LBMK_REQUIRE((n + 31) / 32 < (1LL << 31), "sum_partials: grid");
  if (nseg <= 16 && n % 4 == 0 && n >= 4096 && lbmk_aligned(partial, 16) && lbmk_aligned(out, out_dtype ? 16 : 8)) {
    hipLaunchKernelGGL(sum_partials_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                       out, n / 4, n, nseg, out_dtype);
    return lbmk_check_launch("sum_partials");
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, partial, out, n,
                     nseg, out_dtype);
  return lbmk_check_launch("sum_partials");
}

extern "C" int64_t lbmk_colsum_scratch_floats(int64_t rows, int64_t rows_per_group, int N) {
  if (rows <= 0 || rows_per_group <= 0 || rows % rows_per_group || N <= 0) return -1;
  const int64_t groups = rows / rows_per_group;
  int64_t nseg = 1024 / (groups * ((N + 63) / 64));
  if (nseg < 1) nseg = 1;
  if (nseg > (rows_per_group + 255) / 256) nseg = (rows_per_group + 255) / 256;
  if (nseg > 65535) nseg = 65535;
  return nseg * groups * N;
}

extern "C" int lbmk_colsum(const void* x, void* out, float* scratch, int64_t rows, int64_t rows_per_group, int N,
                           int64_t ld, int out_dtype, void* stream) {
  LBMK_REQUIRE(x && out && scratch, "colsum: null pointer");
  const int64_t nfl = lbmk_colsum_scratch_floats(rows, rows_per_group, N);
  LBMK_REQUIRE(nfl > 0, "colsum: rows=%lld rows_per_group=%lld N=%d", (long long)rows, (long long)rows_per_group, N);
  LBMK_REQUIRE(N % 8 == 0 && ld >= N && ld % 8 == 0 && lbmk_aligned(x, 16), "colsum: N / ld / alignment");
  const int64_t groups = rows / rows_per_group;
  LBMK_REQUIRE(groups <= 65535, "colsum: %lld groups", (long long)groups);
  const int nseg = (int)(nfl / (groups * N));
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)groups, (unsigned)nseg);
  hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scratch,
                     rows_per_group, N, ld, nseg);
  const int64_t GN = groups * N;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((GN + 31) / 32)), dim3(256), 0, (hipStream_t)stream, scratch,
                     out, GN, nseg, out_dtype);
  return lbmk_check_launch("colsum");
}

// ================================================================================================
// Normalisation / activation / softmax / loss backward (HBM-bound)
// ================================================================================================
namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// d/dz silu(z)
__device__ __forceinline__ float dsilu_f(float z) { const float s = sigmoid_f(z); return s * (1.f + z * (1.f - s)); }
// d/dg gelu_erf(g) = Phi(g) + g * phi(g)
__device__ __forceinline__ float dgelu_erf_f(float g) {
  return 0.5f * (1.f + erf_as_f(g * 0.70710678118654752f)) + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}

// ---- GroupNorm(+SiLU) backward -------------------------------------------------------------------
// pass 1: per (image, row slab, channel):  s1 = sum dz,  s2 = sum dz * xhat   (dz = dy * silu'(z) when fused)
// thread mapping as in the forward statistics pass: a thread owns one 8-channel chunk and strides rows.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ partial,
                                                           int HW, int C, int G, int rows_per_slab, int nslab) {
  extern __shared__ float sm[];   // [thread][16]: s1[8], s2[8]; reduced over the row lanes in a fixed order
  const int t = threadIdx.x;
  const int nch = C >> 3;                 // 8-channel chunks per row
  const int cpb = nch < 256 ? nch : 256;  // chunks handled per block pass (grid.y covers the rest)
  const int rpp = 256 / cpb;
  const int chunk = blockIdx.y * cpb + t % cpb, rl = t / cpb;
  const int slab = blockIdx.x, b = blockIdx.z;
  const int cpg = C / G;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (chunk < nch && rl < rpp) {
    const int c0 = chunk * 8;
    float a[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c0 + e) / cpg;
      mu[e] = stats[((int64_t)b * G + g) * 2];
      rs[e] = stats[((int64_t)b * G + g) * 2 + 1];
      a[e] = gamma ? gamma[c0 + e] : 1.f;
      sh[e] = beta ? beta[c0 + e] : 0.f;
    }
    const int r0 = slab * rows_per_slab, r1 = min(r0 + rows_per_slab, HW);
    const int64_t base = ((int64_t)b * HW) * C + c0;
    for (int r = r0 + rl; r < r1; r += rpp) {
      const u16x8 xv = *(const u16x8*)(x + base + (int64_t)r * C);
      const u16x8 dv = *(const u16x8*)(dy + base + (int64_t)r * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (bf2f(xv.v[e]) - mu[e]) * rs[e];
        float dz = bf2f(dv.v[e]);
        if (SILU) dz *= dsilu_f(xh * a[e] + sh[e]);
        s1[e] += dz;
        s2[e] += dz * xh;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sm[t * 16 + e] = s1[e]; sm[t * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  // partial[((b*nslab + slab)*C + c)*2 + {0,1}]: row lanes summed in order (no atomics: bit-reproducible)
  for (int i = t; i < cpb * 8; i += 256) {
    const int ck = i >> 3, e = i & 7;
    const int c = (blockIdx.y * cpb + ck) * 8 + e;
    if (c < C) {
      float a1 = 0.f, a2 = 0.f;
      for (int r = 0; r < rpp; ++r) {
        a1 += sm[(r * cpb + ck) * 16 + e];
        a2 += sm[(r * cpb + ck) * 16 + 8 + e];
      }
      float* p = partial + (((int64_t)b * nslab + slab) * C + c) * 2;
      p[0] = a1;
      p[1] = a2;
    }
  }
}

// pass 2a: bc[b][c][2] = sum over slabs.  32 (image, channel) pairs per workgroup, 8 slab lanes per pair with
// independent loads in flight; the lane sums are added in a fixed order.
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ bc, int B, int C, int nslab) {
  __shared__ double red[8][32][2];
  const int t = threadIdx.x, cl = t & 31, sl = t >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + cl;   // over B*C
  const bool ok = i < (int64_t)B * C;
  double s1 = 0.0, s2 = 0.0;
  if (ok) {
    const int b = (int)(i / C), c = (int)(i - (int64_t)b * C);
    for (int k = sl; k < nslab; k += 8) {
      const float2 p = *(const float2*)(partial + (((int64_t)b * nslab + k) * C + c) * 2);
      s1 += p.x; s2 += p.y;
    }
  }
  red[sl][cl][0] = s1;
  red[sl][cl][1] = s2;
  __syncthreads();
  if (t < 32 && ok) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1 += red[k][t][0]; a2 += red[k][t][1]; }
    bc[i * 2] = (float)a1;
    bc[i * 2 + 1] = (float)a2;
  }
}

// pass 2b: dgamma[c] = sum_b s2, dbeta[c] = sum_b s1;  gsum[b][g] = {sum_c gamma*s1, sum_c gamma*s2} / n
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ bc, const float* __restrict__ gamma,
                                       void* __restrict__ dgamma, void* __restrict__ dbeta, int param_f32, float* __restrict__ gsum,
                                       int B, int C, int G, float inv_n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < C) {
    float a = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) { bsum += bc[((int64_t)b * C + i) * 2]; a += bc[((int64_t)b * C + i) * 2 + 1]; }
    if (param_f32) {
      if (dgamma) ((float*)dgamma)[i] = a;
      if (dbeta) ((float*)dbeta)[i] = bsum;
    } else {
      if (dgamma) ((bf16_t*)dgamma)[i] = f2bf(a);
      if (dbeta) ((bf16_t*)dbeta)[i] = f2bf(bsum);
    }
  }
  if (i < B * G) {
    const int b = i / G, g = i - b * G, cpg = C / G;
    float A = 0.f, Bq = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float ga = gamma ? gamma[c] : 1.f;
      A += ga * bc[((int64_t)b * C + c) * 2];
      Bq += ga * bc[((int64_t)b * C + c) * 2 + 1];
    }
    gsum[i * 2] = A * inv_n;
    gsum[i * 2 + 1] = Bq * inv_n;
  }
}

// pass 3: dx = rstd * (gamma*dz - A/n - xhat * Bq/n)
template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, const float* __restrict__ stats,
                                                           const float* __restrict__ gsum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int HW, int C, int G, int64_t total8,
                                                           const bf16_t* __restrict__ add) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 8-channel chunk of one pixel
  if (gid >= total8) return;
  const int nch = C >> 3;
  const int chunk = (int)(gid % nch);
  const int64_t row = gid / nch;            // b*HW + pixel
  const int b = (int)(row / HW);
  const int c0 = chunk * 8, cpg = C / G;
  const u16x8 xv = *(const u16x8*)(x + row * C + c0);
  const u16x8 dv = *(const u16x8*)(dy + row * C + c0);
  u16x8 av;
  if (add) av = *(const u16x8*)(add + row * C + c0);
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (c0 + e) / cpg;
    const float mu = stats[((int64_t)b * G + g) * 2], rs = stats[((int64_t)b * G + g) * 2 + 1];
    const float A = gsum[((int64_t)b * G + g) * 2], Bq = gsum[((int64_t)b * G + g) * 2 + 1];
    const float ga = gamma ? gamma[c0 + e] : 1.f, be = beta ? beta[c0 + e] : 0.f;
    const float xh = (bf2f(xv.v[e]) - mu) * rs;
    float dz = bf2f(dv.v[e]);
    if (SILU) dz *= dsilu_f(xh * ga + be);
    // the gradient of the tensor's other consumer (a residual link) is added to the ROUNDED gradient of this one, as
    // a separate addition of two bf16 tensors would
    const float v = bf2f(f2bf(rs * (ga * dz - A - xh * Bq)));
    o.v[e] = add ? f2bf(v + bf2f(av.v[e])) : f2bf(v);
  }
  *(u16x8*)(dx + row * C + c0) = o;
}

// ---- LayerNorm backward: one wave per row, a block walks `rows_per_block` rows and keeps dgamma / dbeta partials
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                     bf16_t* __restrict__ dx, const float* __restrict__ gamma,
                                                     float* __restrict__ partial, int rows, int C, float eps,
                                                     int rows_per_block, const bf16_t* __restrict__ add) {
  extern __shared__ float sm[];     // [4 waves][2][C] partials of dgamma, dbeta, summed over the waves in order
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nchunks = C >> 3;
  float dg[NC][8], db[NC][8], ga[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[i][e] = 0.f; db[i][e] = 0.f;
      const int c = (lane + 64 * i) * 8 + e;
      ga[i][e] = (gamma && c < C) ? gamma[c] : 1.f;
    }
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(r_begin + rows_per_block, rows);
  for (int row = r_begin + wave; row < r_end; row += 4) {
    float xv[NC][8], dv[NC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunks) {
        const u16x8 u = *(const u16x8*)(x + (int64_t)row * C + c * 8);
        const u16x8 d = *(const u16x8*)(dy + (int64_t)row * C + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[i][e] = bf2f(u.v[e]); dv[i][e] = bf2f(d.v[e]); s += xv[i][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[i][e] = 0.f; dv[i][e] = 0.f; }
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (lane + 64 * i < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[i][e] - mean; q += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (lane + 64 * i < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[i][e] - mean) * rstd;
          xv[i][e] = xh;
          const float g = ga[i][e] * dv[i][e];
          m1 += g; m2 += g * xh;
          dg[i][e] += dv[i][e] * xh;
          db[i][e] += dv[i][e];
        }
      }
    m1 = wave_sum(m1) / (float)C;
    m2 = wave_sum(m2) / (float)C;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunks) {
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = f2bf(rstd * (ga[i][e] * dv[i][e] - m1 - xv[i][e] * m2));
        if (add) {   // + the gradient that reached the same tensor through its residual link (sum of two bf16 tensors)
          const u16x8 a = *(const u16x8*)(add + (int64_t)row * C + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o.v[e] = f2bf(bf2f(o.v[e]) + bf2f(a.v[e]));
        }
        *(u16x8*)(dx + (int64_t)row * C + c * 8) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sm[wave * 2 * C + c * 8 + e] = dg[i][e];
        sm[wave * 2 * C + C + c * 8 + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int i = t; i < 2 * C; i += 256)
    partial[(int64_t)blockIdx.x * 2 * C + i] = (sm[i] + sm[2 * C + i]) + (sm[4 * C + i] + sm[6 * C + i]);
}

// ---- elementwise ---------------------------------------------------------------------------------
__global__ void silu_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u16x8 xv = ((const u16x8*)x)[i], dv = ((const u16x8*)dy)[i];
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.v[e] = f2bf(bf2f(dv.v[e]) * dsilu_f(bf2f(xv.v[e])));
  ((u16x8*)dx)[i] = o;
}

// h: [rows][2F] (value | gate);  y[r][f] = h[r][f] * gelu(h[r][F+f])
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ y, int64_t rows, int F) {
  const int f8 = F >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * f8) return;
  const int64_t r = i / f8;
  const int c = (int)(i - r * f8) * 8;
  const u16x8 a = *(const u16x8*)(h + r * 2 * F + c), g = *(const u16x8*)(h + r * 2 * F + F + c);
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.v[e] = f2bf(bf2f(a.v[e]) * gelu_erf_f(bf2f(g.v[e])));
  *(u16x8*)(y + r * F + c) = o;
}

__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dh,
                                 int64_t rows, int F) {
  const int f8 = F >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * f8) return;
  const int64_t r = i / f8;
  const int c = (int)(i - r * f8) * 8;
  const u16x8 a = *(const u16x8*)(h + r * 2 * F + c), g = *(const u16x8*)(h + r * 2 * F + F + c);
  const u16x8 d = *(const u16x8*)(dy + r * F + c);
  u16x8 oa, og;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float gv = bf2f(g.v[e]), dv = bf2f(d.v[e]);
    oa.v[e] = f2bf(dv * gelu_erf_f(gv));
    og.v[e] = f2bf(dv * bf2f(a.v[e]) * dgelu_erf_f(gv));
  }
  *(u16x8*)(dh + r * 2 * F + c) = oa;
  *(u16x8*)(dh + r * 2 * F + F + c) = og;
}

// dS[r][j] = scale * P[r][j] * (dP[r][j] - sum_k dP[r][k] P[r][k]);  one block per row; P bf16, dP f32, dS bf16
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const bf16_t* __restrict__ p, const float* __restrict__ dp,
                                                               bf16_t* __restrict__ ds, int cols, float scale) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  const bf16_t* pr = p + row * cols;
  const float* dr = dp + row * cols;
  float s = 0.f;
  for (int j = t * 4; j < cols; j += 1024) {
    const u16x4 pv = *(const u16x4*)(pr + j);
    const f32x4 dv = *(const f32x4*)(dr + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += bf2f(pv.v[e]) * dv[e];
  }
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  for (int j = t * 4; j < cols; j += 1024) {
    const u16x4 pv = *(const u16x4*)(pr + j);
    const f32x4 dv = *(const f32x4*)(dr + j);
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = f2bf(scale * bf2f(pv.v[e]) * (dv[e] - dot));
    *(u16x4*)(ds + row * cols + j) = o;
  }
}

// d pred of the per-sample latent loss: g[b] * 2 (pred - tgt) m^2 / n  (l2)  or  g[b] * sign(pred - tgt) |m| / n  (l1)
template <typename TP, typename TZ>
__global__ void latent_loss_bwd_kernel(const TP* __restrict__ pred, const TZ* __restrict__ z_src, const TZ* __restrict__ z,
                                       const uint8_t* __restrict__ mask, const float* __restrict__ gout, TP* __restrict__ dpred,
                                       int64_t per_sample, int64_t total, int l1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / per_sample);
  const float m = mask ? (mask[i] ? 1.f : 0.f) : 1.f;
  // the forward pass forms target = z_src - z in the latent dtype
  const float tgt = (float)(TZ)((float)z_src[i] - (float)z[i]);
  const float d = ((float)pred[i] - tgt) * m;
  const float g = gout[b] / (float)per_sample;
  const float v = l1 ? g * m * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : g * 2.f * d * m;
  dpred[i] = (TP)v;
}

}  // namespace

// row slabs of the statistics pass: 32 rows each (enough workgroups to fill the chip from 8 images of 32x32 up)
static int gn_bwd_slabs(int HW) {
  const int n = (HW + 31) / 32;
  return n > 256 ? 256 : n;
}

extern "C" int64_t lbmk_groupnorm_bwd_scratch_floats(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C % 8 || G <= 0 || C % G) return -1;
  return (int64_t)B * gn_bwd_slabs(HW) * C * 2 + (int64_t)B * C * 2 + (int64_t)B * G * 2;
}

extern "C" int lbmk_groupnorm_bwd(const void* x, const void* dy, void* dx, const float* stats, const float* gamma,
                                  const float* beta, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C,
                                  int G, int silu, void* stream) {
  return lbmk_groupnorm_bwd_ex(x, dy, dx, stats, gamma, beta, dgamma, dbeta, 1, nullptr, scratch, B, HW, C, G, silu, stream);
}

extern "C" int lbmk_groupnorm_bwd_ex(const void* x, const void* dy, void* dx, const float* stats, const float* gamma,
                                     const float* beta, void* dgamma, void* dbeta, int param_f32, const void* add,
                                     float* scratch, int B, int HW, int C, int G, int silu, void* stream) {
  LBMK_REQUIRE(x && dy && dx && stats && scratch, "groupnorm_bwd: null pointer");
  LBMK_REQUIRE(param_f32 == 0 || param_f32 == 1, "groupnorm_bwd: parameter dtype %d", param_f32);
  LBMK_REQUIRE(!add || lbmk_aligned(add, 16), "groupnorm_bwd: misaligned residual gradient");
  LBMK_REQUIRE(lbmk_groupnorm_bwd_scratch_floats(B, HW, C, G) > 0 && B <= 65535, "groupnorm_bwd: B=%d HW=%d C=%d G=%d", B, HW, C, G);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(dy, 16) && lbmk_aligned(dx, 16), "groupnorm_bwd: misaligned");
  hipStream_t s = (hipStream_t)stream;
  int nslab = gn_bwd_slabs(HW);
  const int rows_per_slab = (HW + nslab - 1) / nslab;
  nslab = (HW + rows_per_slab - 1) / rows_per_slab;
  float* partial = scratch;
  float* bc = scratch + (int64_t)B * gn_bwd_slabs(HW) * C * 2;
  float* gsum = bc + (int64_t)B * C * 2;
  const int nch = C / 8, cpb = nch < 256 ? nch : 256;
  dim3 grid(nslab, (nch + cpb - 1) / cpb, B);
  const size_t shm = (size_t)256 * 16 * sizeof(float);
  if (silu)
    hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, grid, dim3(256), shm, s, (const bf16_t*)x, (const bf16_t*)dy, stats, gamma,
                       beta, partial, HW, C, G, rows_per_slab, nslab);
  else
    hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, grid, dim3(256), shm, s, (const bf16_t*)x, (const bf16_t*)dy, stats, gamma,
                       beta, partial, HW, C, G, rows_per_slab, nslab);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((unsigned)(((int64_t)B * C + 31) / 32)), dim3(256), 0, s, partial, bc, B, C, nslab);
  const int nfin = C > B * G ? C : B * G;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((nfin + 255) / 256), dim3(256), 0, s, bc, gamma, dgamma, dbeta, param_f32, gsum,
                     B, C, G, 1.0f / ((float)HW * (float)(C / G)));
  const int64_t total8 = (int64_t)B * HW * nch;
  LBMK_REQUIRE((total8 + 255) / 256 < (1LL << 31), "groupnorm_bwd: grid");
  if (silu)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, stats, gsum, gamma, beta, HW, C, G, total8, (const bf16_t*)add);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, stats, gsum, gamma, beta, HW, C, G, total8, (const bf16_t*)add);
  return lbmk_check_launch("groupnorm_bwd");
}

extern "C" int64_t lbmk_layernorm_bwd_scratch_floats(int rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8 || C > 2048) return -1;
  int nblk = (rows + 15) / 16;       // 4 rows per wave: enough workgroups to fill the chip at a few thousand rows
  if (nblk > 1024) nblk = 1024;
  return (int64_t)nblk * 2 * C;
}

extern "C" int lbmk_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, float* dgamma_dbeta,
                                  float* scratch, int rows, int C, float eps, void* stream) {
  return lbmk_layernorm_bwd_ex(x, dy, dx, gamma, dgamma_dbeta, dgamma_dbeta ? dgamma_dbeta + C : nullptr, 1, nullptr, scratch,
                               rows, C, eps, stream);
}

extern "C" int lbmk_layernorm_bwd_ex(const void* x, const void* dy, void* dx, const float* gamma, void* dgamma, void* dbeta,
                                     int param_f32, const void* add, float* scratch, int rows, int C, float eps, void* stream) {
  LBMK_REQUIRE(x && dy && dx && scratch, "layernorm_bwd: null pointer");
  LBMK_REQUIRE(param_f32 == 0 || param_f32 == 1, "layernorm_bwd: parameter dtype %d", param_f32);
  LBMK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma and dbeta go together");
  LBMK_REQUIRE(!add || lbmk_aligned(add, 16), "layernorm_bwd: misaligned residual gradient");
  const int64_t nfl = lbmk_layernorm_bwd_scratch_floats(rows, C);
  LBMK_REQUIRE(nfl > 0, "layernorm_bwd: rows=%d C=%d (C %% 8 == 0, C <= 2048)", rows, C);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(dy, 16) && lbmk_aligned(dx, 16), "layernorm_bwd: misaligned");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)(nfl / (2 * C));
  const int rpb = (rows + nblk - 1) / nblk;
  const int nc = (C / 8 + 63) / 64;
  const size_t shm = (size_t)4 * 2 * C * sizeof(float);
#define LN_BWD(NC) hipLaunchKernelGGL(ln_bwd_kernel<NC>, dim3(nblk), dim3(256), shm, s, (const bf16_t*)x, (const bf16_t*)dy, \
                                      (bf16_t*)dx, gamma, scratch, rows, C, eps, rpb, (const bf16_t*)add)
  switch (nc) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    default: LN_BWD(4); break;
  }
#undef LN_BWD
  // block partials are [nblk][2][C]: row 0 dgamma, row 1 dbeta
  if (dgamma)
    hipLaunchKernelGGL(colsum_final_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, s, scratch, dgamma, (int64_t)2 * C, nblk,
                       param_f32, dbeta, (int64_t)C);
  return lbmk_check_launch("layernorm_bwd");
}

extern "C" int lbmk_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  LBMK_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, "silu_bwd: n=%lld", (long long)n);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)dy, (bf16_t*)dx, n / 8);
  return lbmk_check_launch("silu_bwd");
}

extern "C" int lbmk_geglu_fwd(const void* h, void* y, int64_t rows, int F, void* stream) {
  LBMK_REQUIRE(h && y && rows > 0 && F > 0 && F % 8 == 0, "geglu_fwd: rows=%lld F=%d", (long long)rows, F);
  const int64_t n = rows * (F / 8);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                     (bf16_t*)y, rows, F);
  return lbmk_check_launch("geglu_fwd");
}

extern "C" int lbmk_geglu_bwd(const void* h, const void* dy, void* dh, int64_t rows, int F, void* stream) {
  LBMK_REQUIRE(h && dy && dh && rows > 0 && F > 0 && F % 8 == 0, "geglu_bwd: rows=%lld F=%d", (long long)rows, F);
  const int64_t n = rows * (F / 8);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                     (const bf16_t*)dy, (bf16_t*)dh, rows, F);
  return lbmk_check_launch("geglu_bwd");
}

extern "C" int lbmk_softmax_bwd_rows(const void* p, const float* dp, void* ds, int64_t rows, int cols, float scale, void* stream) {
  LBMK_REQUIRE(p && dp && ds && rows > 0 && rows < (1LL << 31) && cols > 0 && cols % 4 == 0, "softmax_bwd_rows: rows/cols");
  LBMK_REQUIRE(lbmk_aligned(p, 8) && lbmk_aligned(dp, 16) && lbmk_aligned(ds, 8), "softmax_bwd_rows: misaligned");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, dp,
                     (bf16_t*)ds, cols, scale);
  return lbmk_check_launch("softmax_bwd_rows");
}

extern "C" int lbmk_latent_loss_bwd(const void* pred, const void* z_src, const void* z, const uint8_t* mask,
                                    const float* gout, void* dpred, int B, int64_t per_sample, int l1, int pred_dtype,
                                    int z_dtype, void* stream) {
  LBMK_REQUIRE(pred && z_src && z && gout && dpred && B > 0 && per_sample > 0, "latent_loss_bwd: arguments");
  const int64_t total = (int64_t)B * per_sample;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define LLB(TP, TZ) hipLaunchKernelGGL((latent_loss_bwd_kernel<TP, TZ>), grid, dim3(256), 0, s, (const TP*)pred, (const TZ*)z_src, \
                                       (const TZ*)z, mask, gout, (TP*)dpred, per_sample, total, l1)
  if (pred_dtype == 0 && z_dtype == 0) LLB(__bf16, __bf16);
  else if (pred_dtype == 0 && z_dtype == 1) LLB(__bf16, float);
  else if (pred_dtype == 1 && z_dtype == 0) LLB(float, __bf16);
  else if (pred_dtype == 1 && z_dtype == 1) LLB(float, float);
  else LBMK_REQUIRE(false, "latent_loss_bwd: dtype codes %d %d", pred_dtype, z_dtype);
#undef LLB
  return lbmk_check_launch("latent_loss_bwd");
}

// ---- AdamW (decoupled weight decay), one fused pass over a flat parameter range ---------------------
// torch.optim.AdamW semantics (the reference's optimizer, src/lbm/trainer/trainer.py:118-121):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// Parameters and gradients are bf16; both moments are kept in f32.
namespace {
__global__ void adamw_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float inv_bc1, float inv_sqrt_bc2, float grad_scale) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  if (i + 8 <= n) {
    u16x8 pv = *(const u16x8*)(p + i);
    const u16x8 gv = *(const u16x8*)(g + i);
    float mm[8], vv[8];
    *(f32x4*)mm = *(const f32x4*)(m + i); *(f32x4*)(mm + 4) = *(const f32x4*)(m + i + 4);
    *(f32x4*)vv = *(const f32x4*)(v + i); *(f32x4*)(vv + 4) = *(const f32x4*)(v + i + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gr = bf2f(gv.v[e]) * grad_scale;
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      float w = bf2f(pv.v[e]) * (1.f - lr * wd);
      w -= lr * inv_bc1 * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
      pv.v[e] = f2bf(w);
    }
    *(u16x8*)(p + i) = pv;
    *(f32x4*)(m + i) = *(f32x4*)mm; *(f32x4*)(m + i + 4) = *(f32x4*)(mm + 4);
    *(f32x4*)(v + i) = *(f32x4*)vv; *(f32x4*)(v + i + 4) = *(f32x4*)(vv + 4);
  } else {
    for (int64_t k = i; k < n; ++k) {
      const float gr = bf2f(g[k]) * grad_scale;
      m[k] = b1 * m[k] + (1.f - b1) * gr;
      v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
      float w = bf2f(p[k]) * (1.f - lr * wd);
      w -= lr * inv_bc1 * m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps);
      p[k] = f2bf(w);
    }
  }
}
}  // namespace

extern "C" int lbmk_adamw(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, float grad_scale, void* stream) {
  LBMK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: arguments");
  LBMK_REQUIRE(lbmk_aligned(p, 16) && lbmk_aligned(g, 16) && lbmk_aligned(m, 16) && lbmk_aligned(v, 16), "adamw: 16-byte alignment");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int64_t nthreads = (n + 7) / 8;
  LBMK_REQUIRE((nthreads + 255) / 256 < (1LL << 31), "adamw: grid");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p,
                     (const bf16_t*)g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), grad_scale);
  return lbmk_check_launch("adamw");
}

// ---- pixel reconstruction loss (src/lbm/models/lbm/lbm_model.py:197-249, l2 / l1 variants) -------------
// per-sample mean of ((c(dec) - target) * mask)^2 or |.|, and its gradient w.r.t. dec; c = clamp to [-1, 1] when
// `clamp` is set (the reference clamps the VAE-decoded prediction only: with no VAE the raw prediction is compared)
namespace {
template <typename TP, typename TT>
__global__ __launch_bounds__(1024) void pixel_loss_kernel(const TP* __restrict__ dec, const TT* __restrict__ tgt,
                                                          const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                          int64_t per_sample, int l1, int clamp) {
  __shared__ float red[16];
  const int64_t base = (int64_t)blockIdx.x * per_sample;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per_sample; i += 1024) {
    const float m = mask ? (mask[base + i] ? 1.f : 0.f) : 1.f;
    float x = (float)dec[base + i];
    if (clamp) x = fminf(fmaxf(x, -1.f), 1.f);
    const float d = (x - (float)tgt[base + i]) * m;
    acc += l1 ? fabsf(d) : d * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    out[blockIdx.x] = s / (float)per_sample;
  }
}

template <typename TP, typename TT>
__global__ void pixel_loss_bwd_kernel(const TP* __restrict__ dec, const TT* __restrict__ tgt, const uint8_t* __restrict__ mask,
                                      const float* __restrict__ gout, TP* __restrict__ ddec, int64_t per_sample, int64_t total,
                                      int l1, int clamp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / per_sample);
  const float m = mask ? (mask[i] ? 1.f : 0.f) : 1.f;
  const float x = (float)dec[i];
  const float inside = (!clamp || (x >= -1.f && x <= 1.f)) ? 1.f : 0.f;   // a clamp passes the gradient inside [-1, 1] only
  const float d = ((clamp ? fminf(fmaxf(x, -1.f), 1.f) : x) - (float)tgt[i]) * m;
  const float g = gout[b] / (float)per_sample * inside;
  ddec[i] = (TP)(l1 ? g * m * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : g * 2.f * d * m);
}
}  // namespace

extern "C" int lbmk_pixel_loss(const void* dec, const void* target, const uint8_t* mask, float* out, int B,
                               int64_t per_sample, int l1, int clamp, int dec_dtype, int tgt_dtype, void* stream) {
  LBMK_REQUIRE(dec && target && out && B > 0 && per_sample > 0, "pixel_loss: arguments");
  hipStream_t s = (hipStream_t)stream;
#define PL(TP, TT) hipLaunchKernelGGL((pixel_loss_kernel<TP, TT>), dim3(B), dim3(1024), 0, s, (const TP*)dec, (const TT*)target, \
                                      mask, out, per_sample, l1, clamp)
  if (dec_dtype == 0 && tgt_dtype == 0) PL(__bf16, __bf16);
  else if (dec_dtype == 0 && tgt_dtype == 1) PL(__bf16, float);
  else if (dec_dtype == 1 && tgt_dtype == 0) PL(float, __bf16);
  else if (dec_dtype == 1 && tgt_dtype == 1) PL(float, float);
  else LBMK_REQUIRE(false, "pixel_loss: dtype codes %d %d", dec_dtype, tgt_dtype);
#undef PL
  return lbmk_check_launch("pixel_loss");
}

extern "C" int lbmk_pixel_loss_bwd(const void* dec, const void* target, const uint8_t* mask, const float* gout,
                                   void* ddec, int B, int64_t per_sample, int l1, int clamp, int dec_dtype,
                                   int tgt_dtype, void* stream) {
  LBMK_REQUIRE(dec && target && gout && ddec && B > 0 && per_sample > 0, "pixel_loss_bwd: arguments");
  const int64_t total = (int64_t)B * per_sample;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define PLB(TP, TT) hipLaunchKernelGGL((pixel_loss_bwd_kernel<TP, TT>), grid, dim3(256), 0, s, (const TP*)dec, (const TT*)target, \
                                       mask, gout, (TP*)ddec, per_sample, total, l1, clamp)
  if (dec_dtype == 0 && tgt_dtype == 0) PLB(__bf16, __bf16);
  else if (dec_dtype == 0 && tgt_dtype == 1) PLB(__bf16, float);
  else if (dec_dtype == 1 && tgt_dtype == 0) PLB(float, __bf16);
  else if (dec_dtype == 1 && tgt_dtype == 1) PLB(float, float);
  else LBMK_REQUIRE(false, "pixel_loss_bwd: dtype codes %d %d", dec_dtype, tgt_dtype);
#undef PLB
  return lbmk_check_launch("pixel_loss_bwd");
}

// ---- convolution weight packing ---------------------------------------------------------------------
// [Cout][Cin][KH][KW] parameter -> the GEMM operand [n_p][Kp] with k = tap * cin_p + c (k_order 0) or
// k = (c / 64) * ntaps * 64 + tap * 64 + c % 64 (k_order 1), zero padded.  swap_flip: the input-gradient
// filter instead (rows = input channels, columns over output channels, taps mirrored).
namespace {
__global__ void pack_conv_weight_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int N, int Cin, int KH,
                                        int KW, int rows, int cols, int cols_p, int n_p, int Kp, int k_order, int swap_flip) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n_p * Kp) return;
  const int r = (int)(i / Kp), k = (int)(i - (int64_t)r * Kp);
  const int ntaps = KH * KW;
  bf16_t v = 0;
  if (r < rows && k < ntaps * cols_p) {
    int tap, c;
    if (k_order) {
      const int chunk = k / (ntaps * 64), rem = k - chunk * ntaps * 64;
      tap = rem >> 6;
      c = chunk * 64 + (rem & 63);
    } else {
      tap = k / cols_p;
      c = k - tap * cols_p;
    }
    if (c < cols) {
      const int kh = tap / KW, kw = tap - kh * KW;
      if (swap_flip) v = w[(((int64_t)c * Cin + r) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];   // w[out c][in r]
      else v = w[(((int64_t)r * Cin + c) * KH + kh) * KW + kw];
    }
  }
  out[i] = v;
}
}  // namespace

extern "C" int lbmk_pack_conv_weight(const void* w, void* out, int N, int Cin, int KH, int KW, int n_p, int Kp,
                                     int k_order, int swap_flip, void* stream) {
  LBMK_REQUIRE(w && out && N > 0 && Cin > 0 && KH > 0 && KW > 0, "pack_conv_weight: arguments");
  const int rows = swap_flip ? Cin : N, cols = swap_flip ? N : Cin;
  const int cols_p = (cols + 7) / 8 * 8;
  LBMK_REQUIRE(n_p >= rows && n_p % 8 == 0 && Kp >= KH * KW * cols_p && Kp % 8 == 0, "pack_conv_weight: n_p=%d Kp=%d", n_p, Kp);
  LBMK_REQUIRE(!k_order || cols_p % 64 == 0, "pack_conv_weight: k_order 1 needs whole 64-channel chunks");
  const int64_t total = (int64_t)n_p * Kp;
  LBMK_REQUIRE((total + 255) / 256 < (1LL << 31), "pack_conv_weight: grid");
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w, (bf16_t*)out, N, Cin, KH, KW, rows, cols, cols_p, n_p, Kp, k_order, swap_flip);
  return lbmk_check_launch("pack_conv_weight");
}

