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
                                                        int64_t R, int C, int64_t ld_in, int64_t ld_out) {
  __shared__ bf16_t tile[64][72];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
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

// out[g][n] (+)= sum over segments; out dtype 0 bf16 / 1 f32
__global__ void colsum_final_kernel(const float* __restrict__ partial, void* __restrict__ out, int64_t GN, int nseg,
                                    int out_dtype) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= GN) return;
  float a = 0.f;
  for (int s = 0; s < nseg; ++s) a += partial[(int64_t)s * GN + i];
  if (out_dtype) ((float*)out)[i] = a;
  else ((bf16_t*)out)[i] = f2bf(a);
}

}  // namespace

extern "C" int lbmk_transpose(const void* in, void* out, int64_t R, int C, int64_t ld_in, int64_t ld_out,
                              void* stream) {
  LBMK_REQUIRE(in && out, "transpose: null pointer");
  LBMK_REQUIRE(R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0, "transpose: R=%lld C=%d must be multiples of 8", (long long)R, C);
  LBMK_REQUIRE(ld_in >= C && ld_out >= R && ld_in % 8 == 0 && ld_out % 8 == 0, "transpose: leading dims");
  LBMK_REQUIRE(lbmk_aligned(in, 16) && lbmk_aligned(out, 16), "transpose: misaligned");
  LBMK_REQUIRE((R + 63) / 64 < (1LL << 31) && (C + 63) / 64 <= 65535, "transpose: grid");
  dim3 grid((unsigned)((R + 63) / 64), (unsigned)((C + 63) / 64));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, R, C,
                     ld_in, ld_out);
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
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((GN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scratch,
                     out, GN, nseg, out_dtype);
  return lbmk_check_launch("colsum");
}
