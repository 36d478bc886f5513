// GroupNorm(+SiLU) and LayerNorm over channels-last bf16 activations (HBM-bound kernels).
//
// GroupNorm: x is [B][HW][C]; a group is C/G consecutive channels of every pixel of one
// image.  Three launches: (1) per-slab partial sums, every thread owning one fixed 8-channel
// chunk and striding over rows with 16-byte loads; (2) a tiny deterministic reduction of the
// slab partials into mean / rstd; (3) apply y = x*a[c] + b[c] (+SiLU) with the same mapping.
// No float atomics anywhere (LDS included): every sum has a fixed order, results are bit-reproducible.
#include "common.h"
#include <mutex>

namespace {

struct GnPlan {
  int S;       // channel segments (grid.y); each holds whole groups
  int nch;     // 8-channel chunks per segment
  int rpp;     // rows handled per pass by one block
  int nslab;   // row slabs per image (grid.x)
  int rows_per_slab;
};

bool gn_plan(int B, int HW, int C, int G, GnPlan* pl) {
  if (C % 8 || G <= 0 || C % G) return false;
  int S = 1;
  while ((C / 8 + S - 1) / S > 256 || (C / 8) % S || G % S || (C / S) % (C / G)) {
    ++S;
    if (S > G) return false;
  }
  pl->S = S;
  pl->nch = C / 8 / S;
  pl->rpp = 256 / pl->nch;
  // slabs per image do NOT depend on the batch size: the order of the partial sums (hence the last bit of mean / rstd,
  // hence every output) must be the same for an image alone and inside a batch
  (void)B;
  int want = (2048 + S - 1) / S;
  int max_slabs = (HW + pl->rpp * 8 - 1) / (pl->rpp * 8);
  if (max_slabs < 1) max_slabs = 1;
  int nslab = want < max_slabs ? want : max_slabs;
  if (nslab < 1) nslab = 1;
  pl->rows_per_slab = (HW + nslab - 1) / nslab;
  pl->nslab = (HW + pl->rows_per_slab - 1) / pl->rows_per_slab;
  return true;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x,
                                                       float* __restrict__ partial, int HW, int C,
                                                       int G, int nch, int rpp, int rows_per_slab,
                                                       int nslab) {
  // fixed-order reduction (no float atomics anywhere: results are bit-reproducible run to run)
  __shared__ float sums[2][256 * 8];   // [sum | sumsq][row lane][channel of the segment]
  __shared__ float chs[2][2048];       // per-channel totals of this block
  const int t = threadIdx.x;
  const int slab = blockIdx.x, seg = blockIdx.y, b = blockIdx.z;
  const int cpg = C / G;
  const int gps = G / gridDim.y;     // groups per segment
  const int cseg = nch * 8;          // channels per segment
  const int chunk = t % nch, rl = t / nch;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  if (rl < rpp) {
    const int r0 = slab * rows_per_slab;
    const int r1 = min(r0 + rows_per_slab, HW);
    const bf16_t* base = x + ((int64_t)b * HW) * C + seg * cseg + chunk * 8;
    auto accum = [&](const u16x8& v) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = bf2f(v.v[e]);
        s[e] += f;
        ss[e] += f * f;
      }
    };
    int r = r0 + rl;
    // four independent 16-byte loads in flight per lane
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      const u16x8 v0 = *(const u16x8*)(base + (int64_t)r * C);
      const u16x8 v1 = *(const u16x8*)(base + (int64_t)(r + rpp) * C);
      const u16x8 v2 = *(const u16x8*)(base + (int64_t)(r + 2 * rpp) * C);
      const u16x8 v3 = *(const u16x8*)(base + (int64_t)(r + 3 * rpp) * C);
      accum(v0); accum(v1); accum(v2); accum(v3);
    }
    for (; r < r1; r += rpp) accum(*(const u16x8*)(base + (int64_t)r * C));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sums[0][rl * cseg + chunk * 8 + e] = s[e];
      sums[1][rl * cseg + chunk * 8 + e] = ss[e];
    }
  }
  __syncthreads();
  // per channel: row lanes in order
  for (int c = t; c < cseg; c += 256) {
    float a = 0.f, a2 = 0.f;
    for (int r = 0; r < rpp; ++r) { a += sums[0][r * cseg + c]; a2 += sums[1][r * cseg + c]; }
    chs[0][c] = a;
    chs[1][c] = a2;
  }
  __syncthreads();
  // per group: its channels in order
  if (t < gps * 2) {
    const int gl = t >> 1, which = t & 1;
    float a = 0.f;
    for (int c = gl * cpg; c < (gl + 1) * cpg; ++c) a += chs[which][c];
    const int g = seg * gps + gl;
    partial[(((int64_t)b * G + g) * nslab + slab) * 2 + which] = a;
  }
}

// one workgroup per (image, group): threads stride over the slab partials, wave + LDS reduction
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial,
                                                          float* __restrict__ stats, int BG, int nslab,
                                                          float inv_count, float eps) {
  __shared__ double red[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = blockIdx.x;
  // accumulate the slab partials in double: E[x^2]-E[x]^2 then loses nothing to the merge
  double s = 0.0, ss = 0.0;
  const float2* pp = (const float2*)partial + (int64_t)i * nslab;
  for (int k = t; k < nslab; k += 256) {
    const float2 v = pp[k];
    s += (double)v.x;
    ss += (double)v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = ss; }
  __syncthreads();
  if (t == 0) {
    s = red[0] + red[2] + red[4] + red[6];
    ss = red[1] + red[3] + red[5] + red[7];
    const double mean = s * inv_count;
    double var = ss * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ y,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW,
                                                       int C, int G, int nch, int rpp,
                                                       int rows_per_slab) {
  const int t = threadIdx.x;
  const int slab = blockIdx.x, seg = blockIdx.y, b = blockIdx.z;
  const int cpg = C / G;
  const int chunk = t % nch, rl = t / nch;
  if (rl >= rpp) return;
  const int c0 = seg * nch * 8 + chunk * 8;
  float a[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    const int g = c / cpg;
    const float mean = stats[((int64_t)b * G + g) * 2], rstd = stats[((int64_t)b * G + g) * 2 + 1];
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    a[e] = rstd * ga;
    sh[e] = be - mean * a[e];
  }
  const int r0 = slab * rows_per_slab;
  const int r1 = min(r0 + rows_per_slab, HW);
  const int64_t base = ((int64_t)b * HW) * C + c0;
  auto apply = [&](const u16x8& v, int64_t row) {
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = bf2f(v.v[e]) * a[e] + sh[e];
      if (SILU) f = silu_f(f);
      o.v[e] = f2bf(f);
    }
    *(u16x8*)(y + base + row * C) = o;
  };
  int r = r0 + rl;
  for (; r + 3 * rpp < r1; r += 4 * rpp) {      // four independent 16-byte loads in flight per lane
    const u16x8 v0 = *(const u16x8*)(x + base + (int64_t)r * C);
    const u16x8 v1 = *(const u16x8*)(x + base + (int64_t)(r + rpp) * C);
    const u16x8 v2 = *(const u16x8*)(x + base + (int64_t)(r + 2 * rpp) * C);
    const u16x8 v3 = *(const u16x8*)(x + base + (int64_t)(r + 3 * rpp) * C);
    apply(v0, r); apply(v1, r + rpp); apply(v2, r + 2 * rpp); apply(v3, r + 3 * rpp);
  }
  for (; r < r1; r += rpp) apply(*(const u16x8*)(x + base + (int64_t)r * C), r);
}

// one wave per row; NC = 8-channel chunks per lane
template <int NC>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x,
                                                        bf16_t* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows,
                                                        int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = C >> 3;
  float v[NC][8];
  float s = 0.f;
  const bf16_t* xr = x + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunks) {
      const u16x8 u = *(const u16x8*)(xr + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(u.v[e]); s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  bf16_t* yr = y + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunks) {
      u16x8 o;
      f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (gamma) { g0 = *(const f32x4*)(gamma + c * 8); g1 = *(const f32x4*)(gamma + c * 8 + 4); }
      if (beta) { b0 = *(const f32x4*)(beta + c * 8); b1 = *(const f32x4*)(beta + c * 8 + 4); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o.v[e] = f2bf((v[i][e] - mean) * rstd * g0[e] + b0[e]);
        o.v[e + 4] = f2bf((v[i][e + 4] - mean) * rstd * g1[e] + b1[e]);
      }
      *(u16x8*)(yr + c * 8) = o;
    }
  }
}

}  // namespace

extern "C" int64_t lbmk_groupnorm_scratch_floats(int B, int HW, int C, int G) {
  GnPlan pl;
  if (B <= 0 || HW <= 0 || !gn_plan(B, HW, C, G, &pl)) return -1;
  return (int64_t)B * G * pl.nslab * 2 + (int64_t)B * G * 2;
}

// the three launches of a GroupNorm pass; scratch = [B*G*nslab*2 slab partials | B*G*2 statistics] floats
static int gn_run(const char* what, const void* x, void* y, const float* gamma, const float* beta, float* scratch,
                  int64_t scratch_cap, int B, int HW, int C, int G, float eps, int silu, void* stream) {
  LBMK_REQUIRE(x && y && scratch, "%s: null pointer", what);
  LBMK_REQUIRE(B > 0 && HW > 0 && B <= 65535, "%s: B=%d HW=%d", what, B, HW);
  GnPlan pl;
  LBMK_REQUIRE(gn_plan(B, HW, C, G, &pl), "%s: unsupported C=%d G=%d", what, C, G);
  LBMK_REQUIRE(G / pl.S <= 32, "%s: more than 32 groups per segment", what);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(y, 16), "%s: x / y must be 16-byte aligned", what);
  const int64_t need = (int64_t)B * G * pl.nslab * 2 + (int64_t)B * G * 2;
  LBMK_REQUIRE(scratch_cap < 0 || need <= scratch_cap,
               "%s: B=%d HW=%d C=%d G=%d needs %lld scratch floats, the built-in buffer holds %lld", what, B, HW, C, G,
               (long long)need, (long long)scratch_cap);
  hipStream_t s = (hipStream_t)stream;
  float* partial = scratch;
  float* stats = scratch + (int64_t)B * G * pl.nslab * 2;
  dim3 grid(pl.nslab, pl.S, B);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, partial, HW, C, G,
                     pl.nch, pl.rpp, pl.rows_per_slab, pl.nslab);
  const int BG = B * G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(BG), dim3(256), 0, s, partial, stats, BG,
                     pl.nslab, 1.0f / ((float)HW * (float)(C / G)), eps);
  if (silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       stats, gamma, beta, HW, C, G, pl.nch, pl.rpp, pl.rows_per_slab);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       stats, gamma, beta, HW, C, G, pl.nch, pl.rpp, pl.rows_per_slab);
  return lbmk_check_launch(what);
}

extern "C" int lbmk_groupnorm(const void* x, void* y, const float* gamma, const float* beta,
                              float* scratch, int B, int HW, int C, int G, float eps, int silu,
                              void* stream) {
  return gn_run("groupnorm", x, y, gamma, beta, scratch, -1, B, HW, C, G, eps, silu, stream);
}

// ---- GroupNorm for callers that cannot allocate: the slab partials and statistics live in a fixed device array ----
// Every kernel of the pass writes each scratch word before any kernel reads it (slab partials by gn_stats_kernel,
// statistics by gn_finalize_kernel), so nothing has to be cleared between calls.
#define LBMK_GN_STATIC_FLOATS (1 << 18)     // 1 MiB per device, held until the process ends
static __device__ __align__(256) float gn_static_scratch[LBMK_GN_STATIC_FLOATS];

// Device address of gn_static_scratch on the current device.  The array's name is only a host-side handle, and each
// GPU holds its own copy at its own address: resolved once per device, with that device current.
static int gn_static_ptr(float** out) {
  enum { MAX_DEV = 64 };
  static float* ptrs[MAX_DEV];
  static std::mutex mu;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    lbmk_set_error("groupnorm_static: hipGetDevice -> %d (%s)", dev, hipGetErrorString(e));
    return LBMK_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(gn_static_scratch));
    if (e != hipSuccess || !p) {
      lbmk_set_error("groupnorm_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return LBMK_ERR_LAUNCH;
    }
    ptrs[dev] = p;
  }
  *out = ptrs[dev];
  return LBMK_OK;
}

extern "C" int64_t lbmk_groupnorm_static_max_floats(void) { return LBMK_GN_STATIC_FLOATS; }

extern "C" int lbmk_groupnorm_static(const void* x, void* y, const float* gamma, const float* beta, int B, int HW,
                                     int C, int G, float eps, int silu, void* stream) {
  float* scratch = nullptr;
  const int rc = gn_static_ptr(&scratch);
  if (rc != LBMK_OK) return rc;
  return gn_run("groupnorm_static", x, y, gamma, beta, scratch, LBMK_GN_STATIC_FLOATS, B, HW, C, G, eps, silu,
                stream);
}

extern "C" int lbmk_groupnorm_from_partials(const void* x, void* y, const float* gamma, const float* beta,
                                            const float* partial, int nslab, float* scratch, int B, int HW,
                                            int C, int G, float eps, int silu, void* stream) {
  LBMK_REQUIRE(x && y && scratch && partial, "groupnorm_from_partials: null pointer");
  LBMK_REQUIRE(B > 0 && HW > 0 && B <= 65535 && nslab > 0, "groupnorm_from_partials: B=%d HW=%d nslab=%d", B, HW, nslab);
  GnPlan pl;
  LBMK_REQUIRE(gn_plan(B, HW, C, G, &pl), "groupnorm: unsupported C=%d G=%d", C, G);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(y, 16) && lbmk_aligned(partial, 8), "groupnorm: misaligned");
  hipStream_t s = (hipStream_t)stream;
  const int BG = B * G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(BG), dim3(256), 0, s, partial, scratch, BG, nslab,
                     1.0f / ((float)HW * (float)(C / G)), eps);
  dim3 grid(pl.nslab, pl.S, B);
  if (silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       scratch, gamma, beta, HW, C, G, pl.nch, pl.rpp, pl.rows_per_slab);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                       scratch, gamma, beta, HW, C, G, pl.nch, pl.rpp, pl.rows_per_slab);
  return lbmk_check_launch("groupnorm_from_partials");
}

extern "C" int lbmk_layernorm(const void* x, void* y, const float* gamma, const float* beta,
                              int rows, int C, float eps, void* stream) {
  LBMK_REQUIRE(x && y, "layernorm: null pointer");
  LBMK_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "layernorm: rows=%d C=%d (C %% 8 == 0, <= 2048)", rows, C);
  LBMK_REQUIRE(lbmk_aligned(x, 16) && lbmk_aligned(y, 16) && lbmk_aligned(gamma, 16) && lbmk_aligned(beta, 16),
               "layernorm: x / y / gamma / beta must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4);
  const int nc = (C / 8 + 63) / 64;
#define LN_LAUNCH(NCV)                                                                         \
  hipLaunchKernelGGL(layernorm_kernel<NCV>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, \
                     gamma, beta, rows, C, eps)
  switch (nc) {
    case 1: LN_LAUNCH(1); break;
    case 2: LN_LAUNCH(2); break;
    case 3: LN_LAUNCH(3); break;
    default: LN_LAUNCH(4); break;
  }
#undef LN_LAUNCH
  return lbmk_check_launch("layernorm");
}
