// Bandwidth-bound kernels of the bridge step: interpolant, Euler step + re-noise, latent loss,
// posterior sampling, layout changes at the NCHW API boundary, skip-connection concat,
// timestep embedding and the Gaussian tile merge.  All coalesced, 8-16 bytes per lane.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void st<float>(float* p, int64_t i, float v) { p[i] = v; }
// value after a round trip through the storage type (the reference materialises intermediates)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }

template <typename TI, typename TO>
__global__ void bridge_interp_kernel(const TI* __restrict__ zs, const TI* __restrict__ z,
                                     const TI* __restrict__ eps, const float* __restrict__ sigma,
                                     const uint8_t* __restrict__ copy_src, TO* __restrict__ out,
                                     int64_t per_sample, float bsig) {
  const int b = blockIdx.y;
  const float s = sigma[b];
  const bool cp = copy_src && copy_src[b];
  const float noise_w = bsig * sqrtf(s * (1.f - s));
  const int64_t base = (int64_t)b * per_sample;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float a = ld(zs, base + i);
    float v;
    if (cp) v = a;
    else {
      // same association order as  s*z_src + (1-s)*z + (sigma_b*sqrt(s(1-s)))*eps
      v = s * a + (1.f - s) * ld(z, base + i);
      if (eps) v += noise_w * ld(eps, base + i);
    }
    st(out, base + i, v);
  }
}

template <typename T>
__global__ void euler_step_kernel(const T* __restrict__ x, const T* __restrict__ pred,
                                  const T* __restrict__ eps, T* __restrict__ out, int64_t n,
                                  float ds, float noise_w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = rnd<T>(ld(x, i) + ds * ld(pred, i));   // scheduler.step: f32 math, cast to pred dtype
    if (eps) v = v + noise_w * ld(eps, i);
    st(out, i, v);
  }
}

template <typename TP, typename T>
__global__ __launch_bounds__(1024) void latent_loss_kernel(const TP* __restrict__ pred,
                                                           const T* __restrict__ zs,
                                                           const T* __restrict__ z,
                                                           const uint8_t* __restrict__ mask,
                                                           float* __restrict__ out,
                                                           int64_t per_sample, int l1) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const int64_t base = (int64_t)b * per_sample;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per_sample; i += blockDim.x) {
    const float mk = mask ? (mask[base + i] ? 1.f : 0.f) : 1.f;
    // the reference forms target = z_src - z in the latent dtype before the loss
    const float tgt = rnd<T>(ld(zs, base + i) - ld(z, base + i));
    const float d = (ld(pred, base + i) - tgt) * mk;
    acc += l1 ? fabsf(d) : d * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    out[b] = s / (float)per_sample;
  }
}

// out[b][i] = x[b][i] + a[b] * y[b][i]   (f32 out; predicted x0 = x_t - sigma * prediction)
template <typename TX, typename TY>
__global__ void axpy_rows_kernel(const TX* __restrict__ x, const TY* __restrict__ y,
                                 const float* __restrict__ a, float* __restrict__ out,
                                 int64_t per_sample) {
  const int b = blockIdx.y;
  const float av = a[b];
  const int64_t base = (int64_t)b * per_sample;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample;
       i += (int64_t)gridDim.x * blockDim.x)
    out[base + i] = ld(x, base + i) + av * ld(y, base + i);
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const T* __restrict__ x, bf16_t* __restrict__ y, int C,
                                    int64_t HW, int Cp) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const T* xb = x + (int64_t)b * C * HW + p;
  bf16_t* yp = y + ((int64_t)b * HW + p) * Cp;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      o.v[e] = c < C ? f2bf(ld(xb, (int64_t)c * HW)) : (bf16_t)0;
    }
    *(u16x8*)(yp + c0) = o;
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, T* __restrict__ y, int C,
                                    int64_t HW, int Cs) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const bf16_t* xp = x + ((int64_t)b * HW + p) * Cs;
  T* yb = y + (int64_t)b * C * HW + p;
  for (int c = 0; c < C; ++c) st(yb, (int64_t)c * HW, bf2f(xp[c]));
}

__global__ void copy_channels_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                     int64_t total_chunks, int nch, int src_cs, int src_off,
                                     int dst_cs, int dst_off) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / nch;
    const int c = (int)(i - p * nch) * 8;
    *(u16x8*)(dst + p * dst_cs + dst_off + c) = *(const u16x8*)(src + p * src_cs + src_off + c);
  }
}

template <typename T>
__global__ void posterior_sample_kernel(const bf16_t* __restrict__ mom, const T* __restrict__ eps,
                                        T* __restrict__ z, int Cz, int64_t HW, int ms, float shift,
                                        float scale) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const bf16_t* mp = mom + ((int64_t)b * HW + p) * ms;
  for (int c = 0; c < Cz; ++c) {
    const float mean = bf2f(mp[c]);
    float logvar = bf2f(mp[Cz + c]);
    logvar = fminf(fmaxf(logvar, -30.f), 20.f);
    const float stdv = rnd<T>(expf(0.5f * logvar));
    const int64_t o = ((int64_t)b * Cz + c) * HW + p;
    const float smp = rnd<T>(mean + rnd<T>(stdv * ld(eps, o)));
    st(z, o, (rnd<T>(smp - shift)) * scale);
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out,
                                          int dim) {
  const int b = blockIdx.x;
  const int half = dim >> 1;
  const float tv = t[b];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float freq = expf(-9.210340371976184f * (float)i / (float)half);
    const float arg = tv * freq;
    out[(int64_t)b * dim + i] = f2bf(cosf(arg));
    out[(int64_t)b * dim + half + i] = f2bf(sinf(arg));
  }
}

template <int MODE>   // 0: silu, 1: x*a + b
__global__ void unary_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t nchunks,
                             float a, float bsh) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    const u16x8 v = *(const u16x8*)(x + i * 8);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = bf2f(v.v[e]);
      o.v[e] = f2bf(MODE == 0 ? silu_f(f) : f * a + bsh);
    }
    *(u16x8*)(y + i * 8) = o;
  }
}

__global__ void tile_accumulate_kernel(const bf16_t* __restrict__ tile, float* __restrict__ acc,
                                       float* __restrict__ wsum, const float* __restrict__ wy,
                                       const float* __restrict__ wx, int C, int Cs, int th_full,
                                       int tw_full, int th, int tw, int H, int W, int y0, int x0) {
  const int b = blockIdx.z;
  const int y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= tw || y >= th) return;
  const int oy = y0 + y, ox = x0 + x;
  if (oy >= H || ox >= W) return;
  const float w = wy[y] * wx[x];
  const bf16_t* tp = tile + (((int64_t)b * th_full + y) * tw_full + x) * Cs;
  for (int c = 0; c < C; ++c)
    acc[(((int64_t)b * C + c) * H + oy) * W + ox] += bf2f(tp[c]) * w;
  if (b == 0) wsum[(int64_t)oy * W + ox] += w;
}

template <typename T>
__global__ void tile_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ wsum,
                                     T* __restrict__ out, int64_t HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    st(out, i, acc[i] / wsum[i % HW]);
}

inline unsigned grid1d(int64_t n, int block, int64_t cap = 4096) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int lbmk_bridge_interp(const void* z_src, const void* z, const void* eps,
                                  const float* sigma, const uint8_t* copy_src, void* out, int B,
                                  int64_t per_sample, float bridge_sigma, int in_dtype,
                                  int out_dtype, void* stream) {
  LBMK_REQUIRE(z_src && z && sigma && out, "bridge_interp: null pointer");
  LBMK_REQUIRE(B > 0 && B <= 65535 && per_sample > 0, "bridge_interp: sizes");
  LBMK_REQUIRE((in_dtype | 1) == 1 && (out_dtype | 1) == 1, "bridge_interp: dtype");
  dim3 grid(grid1d(per_sample, 256, 1024), B);
  hipStream_t s = (hipStream_t)stream;
#define BI(TI, TO)                                                                               \
  hipLaunchKernelGGL((bridge_interp_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)z_src,    \
                     (const TI*)z, (const TI*)eps, sigma, copy_src, (TO*)out, per_sample,        \
                     bridge_sigma)
  if (in_dtype == 0 && out_dtype == 0) BI(bf16_t, bf16_t);
  else if (in_dtype == 0) BI(bf16_t, float);
  else if (out_dtype == 0) BI(float, bf16_t);
  else BI(float, float);
#undef BI
  return lbmk_check_launch("bridge_interp");
}

extern "C" int lbmk_euler_step(const void* x, const void* pred, const void* eps, void* out,
                               int64_t n, float s, float s_next, float bridge_sigma, int dtype,
                               void* stream) {
  LBMK_REQUIRE(x && pred && out && n > 0, "euler_step: null pointer / empty");
  LBMK_REQUIRE((dtype | 1) == 1, "euler_step: dtype");
  const float ds = s_next - s;
  const float nw = bridge_sigma * sqrtf(s_next * (1.f - s_next));
  hipStream_t st_ = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(euler_step_kernel<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st_,
                       (const bf16_t*)x, (const bf16_t*)pred, (const bf16_t*)eps, (bf16_t*)out, n, ds, nw);
  else
    hipLaunchKernelGGL(euler_step_kernel<float>, dim3(grid1d(n, 256)), dim3(256), 0, st_,
                       (const float*)x, (const float*)pred, (const float*)eps, (float*)out, n, ds, nw);
  return lbmk_check_launch("euler_step");
}

extern "C" int lbmk_latent_loss(const void* pred, const void* z_src, const void* z,
                                const uint8_t* mask, float* out, int B, int64_t per_sample, int l1,
                                int pred_dtype, int z_dtype, void* stream) {
  LBMK_REQUIRE(pred && z_src && z && out, "latent_loss: null pointer");
  LBMK_REQUIRE(B > 0 && per_sample > 0 && (pred_dtype | 1) == 1 && (z_dtype | 1) == 1, "latent_loss: sizes / dtype");
  hipStream_t s = (hipStream_t)stream;
#define LL(TP, TZ)                                                                                \
  hipLaunchKernelGGL((latent_loss_kernel<TP, TZ>), dim3(B), dim3(1024), 0, s, (const TP*)pred,    \
                     (const TZ*)z_src, (const TZ*)z, mask, out, per_sample, l1)
  if (pred_dtype == 0 && z_dtype == 0) LL(bf16_t, bf16_t);
  else if (pred_dtype == 0) LL(bf16_t, float);
  else if (z_dtype == 0) LL(float, bf16_t);
  else LL(float, float);
#undef LL
  return lbmk_check_launch("latent_loss");
}

extern "C" int lbmk_axpy_rows(const void* x, const void* y, const float* a, float* out, int B,
                              int64_t per_sample, int x_dtype, int y_dtype, void* stream) {
  LBMK_REQUIRE(x && y && a && out && B > 0 && B <= 65535 && per_sample > 0, "axpy_rows: sizes");
  LBMK_REQUIRE((x_dtype | 1) == 1 && (y_dtype | 1) == 1, "axpy_rows: dtype");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(grid1d(per_sample, 256, 1024), B);
#define AX(TX, TY)                                                                              \
  hipLaunchKernelGGL((axpy_rows_kernel<TX, TY>), grid, dim3(256), 0, s, (const TX*)x, (const TY*)y, a, out, per_sample)
  if (x_dtype == 0 && y_dtype == 0) AX(bf16_t, bf16_t);
  else if (x_dtype == 0) AX(bf16_t, float);
  else if (y_dtype == 0) AX(float, bf16_t);
  else AX(float, float);
#undef AX
  return lbmk_check_launch("axpy_rows");
}

extern "C" int lbmk_nchw_to_nhwc(const void* x, void* y, int B, int C, int H, int W, int Cp,
                                 int in_dtype, void* stream) {
  LBMK_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: sizes");
  LBMK_REQUIRE(Cp >= C && Cp % 8 == 0 && lbmk_aligned(y, 16), "nchw_to_nhwc: Cp=%d must be a multiple of 8 >= C", Cp);
  const int64_t HW = (int64_t)H * W;
  dim3 grid((unsigned)((HW + 255) / 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == 0)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, C, HW, Cp);
  else if (in_dtype == 1)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (bf16_t*)y, C, HW, Cp);
  else LBMK_REQUIRE(false, "nchw_to_nhwc: dtype");
  return lbmk_check_launch("nchw_to_nhwc");
}

extern "C" int lbmk_nhwc_to_nchw(const void* x, void* y, int B, int C, int H, int W, int Cs,
                                 int out_dtype, void* stream) {
  LBMK_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && Cs >= C, "nhwc_to_nchw: sizes");
  const int64_t HW = (int64_t)H * W;
  dim3 grid((unsigned)((HW + 255) / 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == 0)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, C, HW, Cs);
  else if (out_dtype == 1)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const bf16_t*)x, (float*)y, C, HW, Cs);
  else LBMK_REQUIRE(false, "nhwc_to_nchw: dtype");
  return lbmk_check_launch("nhwc_to_nchw");
}

extern "C" int lbmk_copy_channels(const void* src, void* dst, int64_t pixels, int C, int src_cs,
                                  int src_off, int dst_cs, int dst_off, void* stream) {
  LBMK_REQUIRE(src && dst && pixels > 0 && C > 0, "copy_channels: sizes");
  LBMK_REQUIRE(C % 8 == 0 && src_cs % 8 == 0 && src_off % 8 == 0 && dst_cs % 8 == 0 && dst_off % 8 == 0,
               "copy_channels: channel counts / offsets must be multiples of 8");
  LBMK_REQUIRE(src_off + C <= src_cs && dst_off + C <= dst_cs, "copy_channels: window exceeds stride");
  LBMK_REQUIRE(lbmk_aligned(src, 16) && lbmk_aligned(dst, 16), "copy_channels: misaligned");
  const int nch = C / 8;
  const int64_t total = pixels * nch;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid1d(total, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, total, nch, src_cs,
                     src_off, dst_cs, dst_off);
  return lbmk_check_launch("copy_channels");
}

extern "C" int lbmk_posterior_sample(const void* moments, const void* eps, void* z, int B, int Cz,
                                     int HW, int ms, float shift, float scale, int dtype,
                                     void* stream) {
  LBMK_REQUIRE(moments && eps && z && B > 0 && B <= 65535 && Cz > 0 && HW > 0, "posterior_sample: sizes");
  LBMK_REQUIRE(ms >= 2 * Cz && (dtype | 1) == 1, "posterior_sample: ms=%d < 2*Cz or dtype", ms);
  dim3 grid((unsigned)((HW + 255) / 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(posterior_sample_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)moments,
                       (const bf16_t*)eps, (bf16_t*)z, Cz, (int64_t)HW, ms, shift, scale);
  else
    hipLaunchKernelGGL(posterior_sample_kernel<float>, grid, dim3(256), 0, s, (const bf16_t*)moments,
                       (const float*)eps, (float*)z, Cz, (int64_t)HW, ms, shift, scale);
  return lbmk_check_launch("posterior_sample");
}

extern "C" int lbmk_timestep_embedding(const float* t, void* out, int B, int dim, void* stream) {
  LBMK_REQUIRE(t && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: sizes");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, t,
                     (bf16_t*)out, dim);
  return lbmk_check_launch("timestep_embedding");
}

extern "C" int lbmk_silu(const void* x, void* y, int64_t n, void* stream) {
  LBMK_REQUIRE(x && y && n > 0 && n % 8 == 0 && lbmk_aligned(x, 16) && lbmk_aligned(y, 16), "silu: n %% 8 / alignment");
  hipLaunchKernelGGL(unary_kernel<0>, dim3(grid1d(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, n / 8, 0.f, 0.f);
  return lbmk_check_launch("silu");
}

extern "C" int lbmk_scale_shift(const void* x, void* y, int64_t n, float a, float b, void* stream) {
  LBMK_REQUIRE(x && y && n > 0 && n % 8 == 0 && lbmk_aligned(x, 16) && lbmk_aligned(y, 16), "scale_shift: n %% 8 / alignment");
  hipLaunchKernelGGL(unary_kernel<1>, dim3(grid1d(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, n / 8, a, b);
  return lbmk_check_launch("scale_shift");
}

extern "C" int lbmk_tile_accumulate(const void* tile, float* acc, float* wsum, const float* wy,
                                    const float* wx, int B, int C, int Cs, int th_full, int tw_full,
                                    int th, int tw, int H, int W, int y0, int x0, void* stream) {
  LBMK_REQUIRE(tile && acc && wsum && wy && wx, "tile_accumulate: null pointer");
  LBMK_REQUIRE(B > 0 && B <= 65535 && C > 0 && Cs >= C && th > 0 && tw > 0 && th <= th_full && tw <= tw_full &&
                   th <= 65535 && y0 >= 0 && x0 >= 0,
               "tile_accumulate: sizes");
  dim3 grid((tw + 255) / 256, th, B);
  hipLaunchKernelGGL(tile_accumulate_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)tile, acc, wsum, wy, wx, C, Cs, th_full, tw_full, th, tw, H, W, y0, x0);
  return lbmk_check_launch("tile_accumulate");
}

extern "C" int lbmk_tile_finalize(const float* acc, const float* wsum, void* out, int B, int C, int H,
                                  int W, int out_dtype, void* stream) {
  LBMK_REQUIRE(acc && wsum && out && B > 0 && C > 0 && H > 0 && W > 0 && (out_dtype | 1) == 1, "tile_finalize: sizes");
  const int64_t HW = (int64_t)H * W, total = HW * C * B;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == 0)
    hipLaunchKernelGGL(tile_finalize_kernel<bf16_t>, dim3(grid1d(total, 256, 8192)), dim3(256), 0, s, acc, wsum, (bf16_t*)out, HW, total);
  else
    hipLaunchKernelGGL(tile_finalize_kernel<float>, dim3(grid1d(total, 256, 8192)), dim3(256), 0, s, acc, wsum, (float*)out, HW, total);
  return lbmk_check_launch("tile_finalize");
}
