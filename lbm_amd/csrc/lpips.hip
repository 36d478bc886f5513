// Kernels of the LPIPS perceptual distance around the VGG16 convolutions (which are igemm launches):
// input scaling, ReLU backward, 2x2 max pooling and the per-layer head
//     d_l[b] = mean over pixels of  sum_c w_c * (f0_c / (|f0| + eps) - f1_c / (|f1| + eps))^2 .
// All HBM-bound; NHWC bf16 activations, f32 arithmetic, fixed-order reductions (no atomics).
#include "common.h"

namespace {

__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int dt) {
  return dt ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st_elem(void* p, int64_t i, int dt, float v) {
  if (dt) ((float*)p)[i] = v;
  else ((bf16_t*)p)[i] = f2bf(v);
}

// y[b][h][w][0..2] = ((clamp ? clip(x, -1, 1) : x) * mask - shift[c]) * inv_scale[c], channels 3..7 zero
__global__ void lpips_prep_kernel(const void* __restrict__ x, const float* __restrict__ mask, bf16_t* __restrict__ y,
                                  int64_t npix, int64_t hw, int x_dt, int clamp, float s0, float s1, float s2, float i0,
                                  float i1, float i2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int64_t b = i / hw, r = i - b * hw;
  const float m = mask ? mask[i] : 1.f;
  const float sh[3] = {s0, s1, s2}, is[3] = {i0, i1, i2};
  u16x8 o;
#pragma unroll
  for (int c = 0; c < 8; ++c) o.v[c] = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = ld_elem(x, (b * 3 + c) * hw + r, x_dt);
    if (clamp) v = fminf(fmaxf(v, -1.f), 1.f);
    o.v[c] = f2bf((v * m - sh[c]) * is[c]);
  }
  *(u16x8*)(y + i * 8) = o;
}

__global__ void lpips_prep_bwd_kernel(const void* __restrict__ x, const float* __restrict__ mask, const bf16_t* __restrict__ dy,
                                      void* __restrict__ dx, int64_t npix, int64_t hw, int x_dt, int clamp, float i0, float i1,
                                      float i2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int64_t b = i / hw, r = i - b * hw;
  const float m = mask ? mask[i] : 1.f;
  const float is[3] = {i0, i1, i2};
  const u16x8 g = *(const u16x8*)(dy + i * 8);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int64_t k = (b * 3 + c) * hw + r;
    float v = bf2f(g.v[c]) * is[c] * m;
    if (clamp) {
      const float xv = ld_elem(x, k, x_dt);
      if (xv < -1.f || xv > 1.f) v = 0.f;
    }
    st_elem(dx, k, x_dt, v);
  }
}

__global__ void relu_bwd_kernel(const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const u16x8 yv = *(const u16x8*)(y + i * 8);
  u16x8 g = *(const u16x8*)(dy + i * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (!(bf2f(yv.v[e]) > 0.f)) g.v[e] = 0;
  *(u16x8*)(dx + i * 8) = g;
}

// 2x2 / stride 2 max pooling, NHWC, one thread per (output pixel, 8-channel chunk)
__global__ void maxpool2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t total, int OH, int OW, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int nch = C >> 3;
  const int ck = (int)(i % nch);
  int64_t p = i / nch;
  const int ox = (int)(p % OW);
  p /= OW;
  const int oy = (int)(p % OH);
  const int64_t b = p / OH;
  const int W = 2 * OW;
  const bf16_t* src = x + ((b * 2 * OH + 2 * oy) * W + 2 * ox) * C + ck * 8;
  const u16x8 a = *(const u16x8*)src, bq = *(const u16x8*)(src + C);
  const u16x8 c = *(const u16x8*)(src + (int64_t)W * C), d = *(const u16x8*)(src + (int64_t)W * C + C);
  u16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    o.v[e] = f2bf(fmaxf(fmaxf(bf2f(a.v[e]), bf2f(bq.v[e])), fmaxf(bf2f(c.v[e]), bf2f(d.v[e]))));
  *(u16x8*)(y + ((b * OH + oy) * OW + ox) * C + ck * 8) = o;
}

// the gradient goes to the first maximum of the window in scan order (0,0) (0,1) (1,0) (1,1), as torch's saved indices do
__global__ void maxpool2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                    int64_t total, int OH, int OW, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int nch = C >> 3;
  const int ck = (int)(i % nch);
  int64_t p = i / nch;
  const int ox = (int)(p % OW);
  p /= OW;
  const int oy = (int)(p % OH);
  const int64_t b = p / OH;
  const int W = 2 * OW;
  const int64_t base = ((b * 2 * OH + 2 * oy) * W + 2 * ox) * C + ck * 8;
  const int64_t off[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
  u16x8 v[4], o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *(const u16x8*)(x + base + off[k]);
  const u16x8 g = *(const u16x8*)(dy + ((b * OH + oy) * OW + ox) * C + ck * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int best = 0;
    float bv = bf2f(v[0].v[e]);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float f = bf2f(v[k].v[e]);
      if (f > bv) { bv = f; best = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k].v[e] = k == best ? g.v[e] : (bf16_t)0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) *(u16x8*)(dx + base + off[k]) = o[k];
}

constexpr float kLpipsEps = 1e-10f;
constexpr int kHeadPix = 256;   // pixels per workgroup of the head kernels

// sum over the lanes of a group of `width` lanes (power of two <= 64); every lane gets the total
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = 1; o < width; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// partial[chunk][b] = sum over the chunk's pixels of sum_c w_c * keep_c * (n0_c - n1_c)^2
// A pixel is handled by C/8 adjacent lanes (8 channels each); a wave covers 512/C pixels per pass.
__global__ __launch_bounds__(256) void lpips_head_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ f1,
                                                         const float* __restrict__ w, const uint8_t* __restrict__ keep,
                                                         float* __restrict__ partial, int HW, int C) {
  __shared__ float red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int gl = C >> 3;                 // lanes per pixel
  const int ppw = 64 / gl;               // pixels per wave pass
  const int sub = lane / gl, cl = lane % gl;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * kHeadPix, p1 = min(p0 + kHeadPix, HW);
  float wc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wc[e] = w[cl * 8 + e];
  float acc = 0.f;
  for (int pb = p0 + wave * ppw; pb < p1; pb += 4 * ppw) {
    const int pix = pb + sub;
    const bool ok = pix < p1;
    const int64_t at = ((int64_t)b * HW + (ok ? pix : p0)) * C + cl * 8;
    const u16x8 a = *(const u16x8*)(f0 + at), c = *(const u16x8*)(f1 + at);
    float u[8], v[8], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      u[e] = bf2f(a.v[e]); v[e] = bf2f(c.v[e]);
      s0 += u[e] * u[e]; s1 += v[e] * v[e];
    }
    s0 = group_sum(s0, gl);
    s1 = group_sum(s1, gl);
    const float r0 = 1.f / (sqrtf(s0) + kLpipsEps), r1 = 1.f / (sqrtf(s1) + kLpipsEps);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float df = u[e] * r0 - v[e] * r1;
      float term = wc[e] * df * df;
      if (keep) term *= keep[at + e] ? 2.f : 0.f;
      d += term;
    }
    d = group_sum(d, gl);
    if (ok && cl == 0) acc += d;
  }
  acc = group_sum(acc, 64);              // only the group leaders hold non-zero sums
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (t == 0) partial[(int64_t)blockIdx.x * gridDim.y + b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// df0 = d/df0 of (gout[b] / HW) * sum_pix sum_c w_c keep_c (n0_c - n1_c)^2
__global__ __launch_bounds__(256) void lpips_head_bwd_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ f1,
                                                             const float* __restrict__ w, const uint8_t* __restrict__ keep,
                                                             const float* __restrict__ gout, bf16_t* __restrict__ df0,
                                                             int HW, int C) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int gl = C >> 3, ppw = 64 / gl;
  const int sub = lane / gl, cl = lane % gl;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * kHeadPix, p1 = min(p0 + kHeadPix, HW);
  const float coef = 2.f * gout[b] / (float)HW;
  float wc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wc[e] = w[cl * 8 + e];
  for (int pb = p0 + wave * ppw; pb < p1; pb += 4 * ppw) {
    const int pix = pb + sub;
    const bool ok = pix < p1;
    const int64_t at = ((int64_t)b * HW + (ok ? pix : p0)) * C + cl * 8;
    const u16x8 a = *(const u16x8*)(f0 + at), c = *(const u16x8*)(f1 + at);
    float u[8], v[8], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      u[e] = bf2f(a.v[e]); v[e] = bf2f(c.v[e]);
      s0 += u[e] * u[e]; s1 += v[e] * v[e];
    }
    s0 = group_sum(s0, gl);
    s1 = group_sum(s1, gl);
    const float n0 = sqrtf(s0), r0 = 1.f / (n0 + kLpipsEps), r1 = 1.f / (sqrtf(s1) + kLpipsEps);
    float g[8], gu = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      g[e] = coef * wc[e] * (u[e] * r0 - v[e] * r1);
      if (keep) g[e] *= keep[at + e] ? 2.f : 0.f;
      gu += g[e] * u[e];
    }
    gu = group_sum(gu, gl);
    // d n0_j / d u_c = delta_jc / (|u| + eps) - u_j u_c / (|u| (|u| + eps)^2); an all-zero pixel gets no second term
    const float k2 = n0 > 0.f ? gu * r0 * r0 / n0 : 0.f;
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = f2bf(g[e] * r0 - k2 * u[e]);
    if (ok) *(u16x8*)(df0 + at) = o;
  }
}

}  // namespace

extern "C" int lbmk_lpips_prep(const void* x, const float* mask, void* y, int B, int H, int W, int x_dtype, int clamp,
                               const float* shift3, const float* inv_scale3, void* stream) {
  LBMK_REQUIRE(x && y && shift3 && inv_scale3 && B > 0 && H > 0 && W > 0, "lpips_prep: arguments");
  LBMK_REQUIRE(x_dtype == 0 || x_dtype == 1, "lpips_prep: x_dtype %d", x_dtype);
  LBMK_REQUIRE(lbmk_aligned(y, 16), "lpips_prep: misaligned output");
  const int64_t hw = (int64_t)H * W, npix = hw * B;
  LBMK_REQUIRE((npix + 255) / 256 < (1LL << 31), "lpips_prep: grid");
  hipLaunchKernelGGL(lpips_prep_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask,
                     (bf16_t*)y, npix, hw, x_dtype, clamp, shift3[0], shift3[1], shift3[2], inv_scale3[0], inv_scale3[1],
                     inv_scale3[2]);
  return lbmk_check_launch("lpips_prep");
}

extern "C" int lbmk_lpips_prep_bwd(const void* x, const float* mask, const void* dy, void* dx, int B, int H, int W,
                                   int x_dtype, int clamp, const float* inv_scale3, void* stream) {
  LBMK_REQUIRE(x && dy && dx && inv_scale3 && B > 0 && H > 0 && W > 0, "lpips_prep_bwd: arguments");
  LBMK_REQUIRE(x_dtype == 0 || x_dtype == 1, "lpips_prep_bwd: x_dtype %d", x_dtype);
  LBMK_REQUIRE(lbmk_aligned(dy, 16), "lpips_prep_bwd: misaligned gradient");
  const int64_t hw = (int64_t)H * W, npix = hw * B;
  LBMK_REQUIRE((npix + 255) / 256 < (1LL << 31), "lpips_prep_bwd: grid");
  hipLaunchKernelGGL(lpips_prep_bwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask,
                     (const bf16_t*)dy, dx, npix, hw, x_dtype, clamp, inv_scale3[0], inv_scale3[1], inv_scale3[2]);
  return lbmk_check_launch("lpips_prep_bwd");
}

extern "C" int lbmk_relu_bwd(const void* y, const void* dy, void* dx, int64_t n, void* stream) {
  LBMK_REQUIRE(y && dy && dx && n > 0 && n % 8 == 0, "relu_bwd: n=%lld", (long long)n);
  LBMK_REQUIRE(lbmk_aligned(y, 16) && lbmk_aligned(dy, 16) && lbmk_aligned(dx, 16), "relu_bwd: misaligned");
  LBMK_REQUIRE((n / 8 + 255) / 256 < (1LL << 31), "relu_bwd: grid");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, n / 8);
  return lbmk_check_launch("relu_bwd");
}

static bool pool_ok(int B, int H, int W, int C) { return B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0; }

extern "C" int lbmk_maxpool2(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  LBMK_REQUIRE(x && y && pool_ok(B, H, W, C), "maxpool2: B=%d H=%d W=%d C=%d (even H, W; C %% 8 == 0)", B, H, W, C);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(y, 16), "maxpool2: misaligned");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  LBMK_REQUIRE((total + 255) / 256 < (1LL << 31), "maxpool2: grid");
  hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, total, H / 2, W / 2, C);
  return lbmk_check_launch("maxpool2");
}

extern "C" int lbmk_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  LBMK_REQUIRE(x && dy && dx && pool_ok(B, H, W, C), "maxpool2_bwd: B=%d H=%d W=%d C=%d", B, H, W, C);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(dy, 16) && lbmk_aligned(dx, 16), "maxpool2_bwd: misaligned");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  LBMK_REQUIRE((total + 255) / 256 < (1LL << 31), "maxpool2_bwd: grid");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, total, H / 2, W / 2, C);
  return lbmk_check_launch("maxpool2_bwd");
}

static bool head_ok(int B, int HW, int C) { return B > 0 && B <= 65535 && HW > 0 && (C == 64 || C == 128 || C == 256 || C == 512); }

extern "C" int64_t lbmk_lpips_head_partial_floats(int B, int HW, int C) {
  if (!head_ok(B, HW, C)) return -1;
  return (int64_t)((HW + kHeadPix - 1) / kHeadPix) * B;
}

extern "C" int lbmk_lpips_head(const void* f0, const void* f1, const float* w, const uint8_t* keep, float* partial,
                               int B, int HW, int C, void* stream) {
  LBMK_REQUIRE(f0 && f1 && w && partial && head_ok(B, HW, C), "lpips_head: B=%d HW=%d C=%d (C in 64/128/256/512)", B, HW, C);
  LBMK_REQUIRE(lbmk_aligned(f0, 16) && lbmk_aligned(f1, 16), "lpips_head: misaligned");
  hipLaunchKernelGGL(lpips_head_kernel, dim3((HW + kHeadPix - 1) / kHeadPix, B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)f0, (const bf16_t*)f1, w, keep, partial, HW, C);
  return lbmk_check_launch("lpips_head");
}

extern "C" int lbmk_lpips_head_bwd(const void* f0, const void* f1, const float* w, const uint8_t* keep, const float* gout,
                                   void* df0, int B, int HW, int C, void* stream) {
  LBMK_REQUIRE(f0 && f1 && w && gout && df0 && head_ok(B, HW, C), "lpips_head_bwd: B=%d HW=%d C=%d", B, HW, C);
  LBMK_REQUIRE(lbmk_aligned(f0, 16) && lbmk_aligned(f1, 16) && lbmk_aligned(df0, 16), "lpips_head_bwd: misaligned");
  hipLaunchKernelGGL(lpips_head_bwd_kernel, dim3((HW + kHeadPix - 1) / kHeadPix, B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)f0, (const bf16_t*)f1, w, keep, gout, (bf16_t*)df0, HW, C);
  return lbmk_check_launch("lpips_head_bwd");
}
