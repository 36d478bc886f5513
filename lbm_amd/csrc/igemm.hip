// Implicit-GEMM convolution / linear layer on bf16 MFMA (gfx950).
//
//   out[m][n] = act(alpha * sum_k A(m,k) * w[n][k] + bias + rowvec + residual)
//
// A(m,k) is gathered on the fly from an NHWC activation tensor (m = output pixel,
// k = (kh, kw, c)), so a 3x3 / 1x1 / stride-2 / nearest-x2-upsampled convolution and a
// plain linear layer are the same kernel.  Both operands are K-contiguous, so both tiles
// are staged global -> LDS with 16-byte LDS-DMA loads (buffer_load_dwordx4 ... lds): the LDS
// image is lane-linear, and the bank-conflict swizzle is applied on the per-lane SOURCE
// offset and again on the ds_read_b128 address.  Out-of-image taps, rows past M / N and
// the K tail use an offset past the end of the buffer descriptor, which reads as zeros.
//
// MFMA orientation: the weight fragment is the A operand and the activation fragment the
// B operand of v_mfma_f32_16x16x32_bf16, so each lane ends up with 4 consecutive output
// channels of one pixel (8-byte bf16 stores along the NHWC channel axis).
#include "common.h"
#include <stdlib.h>

// Diagnostic builds of the wide tiles (STAMP = true, tile ids 115 ... 118) record six 100 MHz
// wall-clock stamps per workgroup: start, after set-up, first stage landed, K loop done, epilogue
// issued, stores drained; slots 6 / 7 hold the shader-clock counter at the start / end of the K loop
// (in-kernel clock = delta(6..7) / delta(2..3) * 100 MHz).  lbmk_debug_igemm_times() copies them out.
// No stamp executes otherwise.
// Only in builds made with -DLBMK_DIAG (make DIAG=1): the product library carries neither the 4 MB stamp array nor the
// synchronising read-out.
#ifdef LBMK_DIAG
__device__ unsigned long long igemm_dbg[65536 * 8];
#define DBG_T(i) do { if (STAMP && threadIdx.x == 0 && blockIdx.x < 65536) igemm_dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DBG_C(i) do { if (STAMP && threadIdx.x == 0 && blockIdx.x < 65536) igemm_dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DBG_T(i) do { } while (0)
#define DBG_C(i) do { } while (0)
#endif


typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ bool lbmk_aligned_dev(const void* p, unsigned a) { return ((uintptr_t)p & (a - 1)) == 0; }

// (image, row, column) of flattened output pixel m.  Image sizes that are powers of two (the shipped models at
// power-of-two resolutions, and every linear layer: one "pixel" per image) take shifts; the general case divides,
// ~25 vector instructions per quotient, which is what a tile's set-up and epilogue were spending their issue slots on.
struct PixelIndex {
  int ohw, ow, sh_ohw, sh_ow;
  bool pow2;
  __device__ __forceinline__ PixelIndex(int oh, int ow_) : ohw(oh * ow_), ow(ow_) {
    pow2 = (ohw & (ohw - 1)) == 0 && (ow & (ow - 1)) == 0;
    sh_ohw = __builtin_ctz(ohw);
    sh_ow = __builtin_ctz(ow);
  }
  __device__ __forceinline__ int image(int m) const { return pow2 ? m >> sh_ohw : m / ohw; }
  __device__ __forceinline__ void operator()(int m, int& b, int& oy, int& ox) const {
    if (pow2) {
      b = m >> sh_ohw;
      const int r = m & (ohw - 1);
      oy = r >> sh_ow;
      ox = r & (ow - 1);
    } else {
      b = m / ohw;
      const int r = m - b * ohw;
      oy = r / ow;
      ox = r - oy * ow;
    }
  }
};

// v + (v of lane ^ 1) + ... summed as the xor butterfly over a 16-lane row sums it, on the vector ALU's lane
// permutes: quad swaps for ^ 1 and ^ 2; then mirrors inside 8 and 16 lanes, which pair every lane with a lane of the
// other quad / other half -- all lanes of a quad (half) hold the same partial sum by then and x + y == y + x, so the
// bits equal the butterfly's.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xb1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4e>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
// v + (v of lane ^ 16) / (v of lane ^ 32): the row swaps leave one register with the even rows' (low half's) values
// on every lane and one with the odd rows' (high half's)
__device__ __forceinline__ float xor16_sum(float v) {
  // (the builtin returns one register for both halves when its operands hold the same value)
  unsigned x = __builtin_bit_cast(unsigned, v), y = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float xor32_sum(float v) {
  // (the builtin returns one register for both halves when its operands hold the same value)
  unsigned x = __builtin_bit_cast(unsigned, v), y = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

// Epilogue shared by every tile shape: acc[i][j] is the 16x16 tile (i, j) of this wave, whose first
// output row / packed column are mbase / nbase.
// Loads are batched ahead of the stores on purpose: `out` may alias anything as far as the compiler
// knows, so a load written after a store waits for it.  The bias of all NI column groups is read once,
// and the residual / rowvec operands of row group i+1 are in flight while row group i is stored.
template <int MI, int NI, bool GNS>
__device__ __forceinline__ void igemm_epilogue(const lbmk_igemm_desc& p, f32x4 (&acc)[MI][NI], int mbase,
                                               int nbase, int lane, int64_t zo, int64_t zr) {
  const int lr = lane & 15, lq = lane >> 4;
  const int ohw = p.OH * p.OW;
  const PixelIndex pix(p.OH, p.OW);
  const bf16_t* __restrict__ res = p.residual ? (const bf16_t*)p.residual + zr : nullptr;
  const bf16_t* __restrict__ rowvec = (const bf16_t*)p.rowvec;
  const float* __restrict__ bias = p.bias;
  const bool geglu = p.act == LBMK_ACT_GEGLU;
  const int n_out = geglu ? (p.N >> 1) : p.N;
  const bool vec_ok = ((n_out & 3) == 0) && ((p.ldo & 3) == 0) &&
                      (!res || (p.ldr & 3) == 0) && (!rowvec || (p.rowvec_ld & 3) == 0);

  // fold alpha and the bias into the accumulators in place (the bias registers die right here)
  {
    float bv[NI][4];
    const bool bias_n = bias && !p.bias_per_m;
    if (bias_n && (p.N & 3) == 0 && lbmk_aligned_dev(bias, 16)) {   // one 16-byte load per column group
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = nbase + j * 16 + lq * 4;
        const f32x4 b4 = n < p.N ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = b4[e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = nbase + j * 16 + lq * 4 + e;
          bv[j][e] = (bias_n && n < p.N) ? bias[n] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = mbase + i * 16 + lr;
      const float bm = (bias && p.bias_per_m && m < p.M) ? bias[m] : 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] * p.alpha + bm + bv[j][e];
    }
  }

  // output row of GEMM row m: itself, or (out_up2) pixel (2y + py, 2x + px) of the twice-as-large output image
  auto orow = [&](int m) -> int64_t {
    if (!p.out_up2) return m;
    int b, y, x;
    pix(m, b, y, x);
    const int par = p.out_up2 - 1;
    return ((int64_t)(b * 2 * p.OH + 2 * y + (par >> 1)) * (2 * p.OW) + 2 * x + (par & 1));
  };
  // output column of element 0 of column group j (GEGLU: value / gate groups alternate, j even is kept)
  auto out_col = [&](int j) { return geglu ? ((nbase + j * 16) >> 1) + lq * 4 : nbase + j * 16 + lq * 4; };

  if (vec_ok) {
    const bool has_add = res || rowvec;
    // GroupNorm partial sums of this wave's 64-row strip (bf16-rounded values, as a separate
    // statistics pass over the stored tensor would read them)
    constexpr bool gn = GNS && MI == 4;   // separate instantiation: the 8 extra accumulators do not fit the plain kernels' budget
    // Wide form (bf16 output, 16-byte aligned rows; an odd last column group keeps its 8-byte accesses): lanes l and l ^ 16 hold
    // channels 4 lq ... + 3 of the same pixel for every column group.  v_permlane16_swap between the packed values of
    // groups j and j + 1 leaves the even-lq lane with 8 consecutive channels of group j and the odd-lq lane with 8 of
    // group j + 1, so a store is 16 bytes per lane and 64 contiguous bytes per pixel: half the store instructions.
    // The residual is read the same way and swapped back into the accumulator layout.
    // (instantiations that reduce GroupNorm statistics take the wide form only: the host checks its conditions)
    const bool wide = (GNS && NI % 2 == 0) ||
                      ((NI >= 2) && !geglu && !p.out_f32 && (n_out & 7) == 0 && (p.ldo & 7) == 0 && (zo & 7) == 0 &&
                       lbmk_aligned_dev(p.out, 16) && (!res || ((p.ldr & 7) == 0 && (zr & 7) == 0 && lbmk_aligned_dev(p.residual, 16))));
    // statistics of the STORED (rounded) values, two per packed dword: sum and sum of squares on the bf16 dot product
    float gsum[NI], gsq[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) gsum[j] = gsq[j] = 0.f;
    auto gn_add = [&](int j, unsigned packed) __attribute__((always_inline)) {
      const bf16x2_t v = __builtin_bit_cast(bf16x2_t, packed);
      gsum[j] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), gsum[j], false);
      gsq[j] = __builtin_amdgcn_fdot2_f32_bf16(v, v, gsq[j], false);
    };
    if (wide) {
      constexpr int NP2 = NI / 2 > 0 ? NI / 2 : 1;
      u32x4 rw[NP2];
      u16x4 rv[NI];
      u16x4 rlast = {{0, 0, 0, 0}};   // residual of an odd last column group
      // first channel of this lane's 16-byte segment of column-group pair jp
      auto wcol = [&](int jp) { return nbase + (2 * jp + (lq & 1)) * 16 + (lq & 2) * 4; };
      auto fetchw = [&](int i) {
        const int m = mbase + i * 16 + lr;
        const bool row_ok = m < p.M;
        const int bimg = (rowvec && row_ok) ? pix.image(m) : 0;
#pragma unroll
        for (int jp = 0; jp < NI / 2; ++jp) {
          const int c = wcol(jp);
          rw[jp] = (res && row_ok && c < n_out) ? *(const u32x4*)(res + (int64_t)m * p.ldr + c) : u32x4{0u, 0u, 0u, 0u};
        }
        if constexpr (NI & 1) {
          const int no = out_col(NI - 1);
          rlast = (res && row_ok && no < n_out) ? *(const u16x4*)(res + (int64_t)m * p.ldr + no) : u16x4{{0, 0, 0, 0}};
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int no = out_col(j);
          rv[j] = (rowvec && row_ok && no < n_out) ? *(const u16x4*)(rowvec + (int64_t)bimg * p.rowvec_ld + no) : u16x4{{0, 0, 0, 0}};
        }
      };
      if (has_add) fetchw(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (has_add) {
#pragma unroll
          for (int jp = 0; jp < NI / 2; ++jp) {
            const auto t0 = __builtin_amdgcn_permlane16_swap(rw[jp][0], rw[jp][2], false, false);
            const auto t1 = __builtin_amdgcn_permlane16_swap(rw[jp][1], rw[jp][3], false, false);
            const unsigned ra[2][2] = {{t0[0], t1[0]}, {t0[1], t1[1]}};   // [group j / j + 1][channel pair]
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned w = ra[jj][e >> 1];
                acc[i][2 * jp + jj][e] += __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16)) + bf2f(rv[2 * jp + jj].v[e]);
              }
          }
          if constexpr (NI & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][NI - 1][e] += bf2f(rlast.v[e]) + bf2f(rv[NI - 1].v[e]);
          }
          if (i + 1 < MI) fetchw(i + 1);
        }
        const int m = mbase + i * 16 + lr;
        const bool row_ok = m < p.M;
        // the activation as one uniform branch per row group (written per element, every element gets its own
        // scalar compare-and-branch pair and the epilogue is bound by instruction issue, not by its stores)
        if (p.act == LBMK_ACT_SILU) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = silu_f(acc[i][j][e]);
        } else if (p.act == LBMK_ACT_RELU) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
        }
        bf16_t* const orow_p = (bf16_t*)p.out + zo + orow(m) * p.ldo;
#pragma unroll
        for (int jp = 0; jp < NI / 2; ++jp) {
          unsigned pk[2][2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jp + jj;
            bf16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][e]);
            pk[jj][0] = (unsigned)o[0] | ((unsigned)o[1] << 16);
            pk[jj][1] = (unsigned)o[2] | ((unsigned)o[3] << 16);
            if (gn) { gn_add(j, pk[jj][0]); gn_add(j, pk[jj][1]); }
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          const int c = wcol(jp);
          if (row_ok && c < n_out) *(u32x4*)(orow_p + c) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
        if constexpr (NI & 1) {
          const int no = out_col(NI - 1);
          u16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(acc[i][NI - 1][e]);
          if (row_ok && no < n_out) *(u16x4*)(orow_p + no) = ov;
        }
      }
    } else {
    u16x4 rr[NI], rv[NI];
    auto fetch = [&](int i) {
      const int m = mbase + i * 16 + lr;
      const bool row_ok = m < p.M;
      const int bimg = (rowvec && row_ok) ? pix.image(m) : 0;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int no = out_col(j);
        const bool ok = row_ok && no < n_out;
        rr[j] = (res && ok) ? *(const u16x4*)(res + (int64_t)m * p.ldr + no) : u16x4{{0, 0, 0, 0}};
        rv[j] = (rowvec && ok) ? *(const u16x4*)(rowvec + (int64_t)bimg * p.rowvec_ld + no) : u16x4{{0, 0, 0, 0}};
      }
    };
    if (has_add) fetch(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (has_add) {
        // consume the operands of row group i in place, then put the loads of row group i+1 in
        // flight BEFORE the stores of row group i
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] += bf2f(rr[j].v[e]) + bf2f(rv[j].v[e]);
        if (i + 1 < MI) fetch(i + 1);
      }
      const int m = mbase + i * 16 + lr;
      if (m >= p.M) continue;
      // activation: one uniform branch per row group (per element it costs a scalar compare-and-branch pair each)
      if (geglu) {
#pragma unroll
        for (int j = 0; j + 1 < NI; j += 2)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] *= gelu_erf_f(acc[i][j + 1][e]);
      } else if (p.act == LBMK_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = silu_f(acc[i][j][e]);
      } else if (p.act == LBMK_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
      }
      const int64_t orow_off = zo + orow(m) * p.ldo;
      if (p.out_f32) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (geglu && (j & 1)) continue;
          const int no = out_col(j);
          if (no < n_out) *(f32x4*)((float*)p.out + orow_off + no) = acc[i][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (geglu && (j & 1)) continue;
          const int no = out_col(j);
          if (no >= n_out) continue;
          u16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(acc[i][j][e]);
          *(u16x4*)((bf16_t*)p.out + orow_off + no) = ov;
          if (gn) {
            gn_add(j, (unsigned)ov.v[0] | ((unsigned)ov.v[1] << 16));
            gn_add(j, (unsigned)ov.v[2] | ((unsigned)ov.v[3] << 16));
          }
        }
      }
    }
    }
    if (gn && mbase < p.M) {
      const int cpg = p.N / p.gn_groups;        // 4, 8 or 16 channels: whole groups inside a 16-column tile
      const int bimg = pix.image(mbase);
      const int strip = ((mbase - bimg * ohw) >> 6) + p.gn_strip_off, nslab = p.gn_nslab > 0 ? p.gn_nslab : ohw >> 6;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        // every row of the strip is a pixel: M is a multiple of 64 whenever statistics are reduced (host check)
        float a = gsum[j], b2 = gsq[j];
        // over the 16 rows held by lanes of equal lq, then over lq: the butterfly lane ^ 1, 2, 4, 8, 16, 32 without the
        // LDS crossbar (a ds_bpermute and its wait per step made this section as long as the stores)
        a = row16_sum(a);
        b2 = row16_sum(b2);
        if (cpg >= 8) { a = xor16_sum(a); b2 = xor16_sum(b2); }
        if (cpg == 16) { a = xor32_sum(a); b2 = xor32_sum(b2); }
        const int col = nbase + j * 16 + lq * 4;
        const bool writer = lr == 0 && col < p.N && (cpg == 4 || (cpg == 8 && !(lq & 1)) || (cpg == 16 && lq == 0));
        if (writer) {
          const int g = col / cpg;
          float* dst = p.gn_partial + (((int64_t)bimg * p.gn_groups + g) * nslab + strip) * 2;
          *(float2*)dst = make_float2(a, b2);
        }
      }
    }
    return;
  }

  // element-wise path for shapes whose rows are not 8-byte aligned (never the case for packed layers)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mbase + i * 16 + lr;
    if (m >= p.M) continue;
    const int bimg = rowvec ? pix.image(m) : 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (geglu && (j & 1)) continue;
      const int no = out_col(j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (no + e >= n_out) break;
        float val = acc[i][j][e];
        if (geglu) val *= gelu_erf_f(acc[i][(j + 1) % NI][e]);
        if (rowvec) val += bf2f(rowvec[(int64_t)bimg * p.rowvec_ld + no + e]);
        if (res) val += bf2f(res[(int64_t)m * p.ldr + no + e]);
        if (p.act == LBMK_ACT_SILU) val = silu_f(val);
        if (p.act == LBMK_ACT_RELU) val = fmaxf(val, 0.f);
        if (p.out_f32) ((float*)p.out)[zo + orow(m) * p.ldo + no + e] = val;
        else ((bf16_t*)p.out)[zo + orow(m) * p.ldo + no + e] = f2bf(val);
      }
    }
  }
}

// Epilogues of the two-pass row softmax over a scores GEMM (attention with a head too wide for the fused kernel):
// SMX = 1 writes, per row and per 64-column wave strip, the maximum of alpha * acc (no tile is stored);
// SMX = 2 stores p = exp2(alpha * acc - smx_max[row]) as bf16 and the strip's sum of the stored (rounded) values.
// Partial index: strip = tile_n * WAVES_N + wn of `parts` strips per row; columns past N take no part.
template <int MI, int NI, int SMX>
__device__ __forceinline__ void igemm_epilogue_smx(const lbmk_igemm_desc& p, f32x4 (&acc)[MI][NI], int mbase, int nbase,
                                                   int lane, int64_t zo, int z, int strip, int parts) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mbase + i * 16 + lr;
    const bool row_ok = m < p.M;
    const int64_t row = (int64_t)z * p.M + (row_ok ? m : 0);
    const float mx = (SMX == 2) ? p.smx_max[row] : 0.f;
    float red = SMX == 1 ? -INFINITY : 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = nbase + j * 16 + lq * 4;
      u16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = col + e < p.N;
        const float sv = acc[i][j][e] * p.alpha;
        if (SMX == 1) red = ok ? fmaxf(red, sv) : red;
        else {
          ov.v[e] = f2bf(ok ? __builtin_amdgcn_exp2f(sv - mx) : 0.f);
          red += bf2f(ov.v[e]);
        }
      }
      if (SMX == 2 && row_ok && col < p.N) *(u16x4*)((bf16_t*)p.out + zo + (int64_t)m * p.ldo + col) = ov;
    }
    if (SMX == 1) { red = fmaxf(red, __shfl_xor(red, 16, 64)); red = fmaxf(red, __shfl_xor(red, 32, 64)); }
    else { red += __shfl_xor(red, 16, 64); red += __shfl_xor(red, 32, 64); }
    if (lq == 0 && row_ok) p.smx_part[row * parts + strip] = red;
  }
}

// FAST: Cin % 64 == 0 -- every lane of a K-tile is on the same filter tap, so the
// tap / channel walk is scalar and a load costs three vector ALU instructions.
// OCC: workgroups meant to share a CU (caps the register budget at 512 / (OCC * waves per SIMD)).
// ROWA (3x3, stride 1, 'same' padding, Cin % 64 == 0): the K loop is bound by the bytes moved L2 -> LDS, and
// the three taps of one filter row read the SAME input pixels shifted by one.  So the activation tile of a
// (filter row, 64-channel chunk) pair is staged once as BM + 8 consecutive pixels of the flattened NHWC input and
// the taps kw = 0, 1, 2 read it at row offsets 0, 1, 2: a third of the activation traffic.  Taps that fall outside
// the image are zeroed on the fragment after the LDS read (per-lane 9-bit masks; waves without such pixels skip it).
// PP (8 waves, Cin % 64 == 0, 1x1 or 2x2 / 3x3 with stride 1 and a same-size output): ping-pong schedule.  Waves 0-3 and 4-7 (one of
// each group per SIMD) run half a K-tile apart: while one group reads its fragments of a K-tile from LDS into registers,
// the other issues the MFMAs of the tile it read in the previous phase, so the matrix pipe of every SIMD always has one
// wave feeding it and no wave's LDS latency is exposed.  An LDS-DMA instruction blocks its wave for on the order of a
// hundred cycles, so every wave issues its loads in its READ phase only, behind its fragment reads.  For that to meet the
// landing deadlines the activation tile has two stage buffers whose halves are staged by the group that does not read
// them, and the weight tile (read by both groups) has three.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DBUF, bool FAST, bool STAMP = false, int OCC = 1, bool GNS = false, bool ROWA = false, bool PP = false, int SMX = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N * OCC + 3) / 4) void igemm_kernel(const lbmk_igemm_desc p) {
  constexpr int NW = WAVES_M * WAVES_N;         // waves per workgroup (4 or 8)
  constexpr int NT = 64 * NW;
  constexpr int RPP = PP ? 32 : NT / 8;         // tile rows staged by one LDS-DMA instruction of the block (PP: of waves 0-3)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int AI = PP ? BM / 64 : BM / RPP;   // LDS-DMA loads per thread for the A tile (PP: for the other group's half)
  constexpr int BI = (BN + RPP - 1) / RPP;      // LDS-DMA loads per thread for the B tile (last one partial when BN % RPP != 0; PP: shared by the groups)
  constexpr int STAGE = (BM + BN) * 128;        // bytes of one K-tile (64 deep) of both operands
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % RPP == 0, "tile shape");
  // all LDS in ONE array: a second __shared__ object makes hipcc drain vmcnt before every ds_read
  constexpr int ABYTES = (BM + 8) * 128;        // ROWA: one activation ring slot (BM + 8 pixels x 64 channels)
  __shared__ __attribute__((aligned(16))) char smem[ROWA ? 2 * ABYTES + 2 * BN * 128 : PP ? 2 * BM * 128 + 3 * BN * 128 : (DBUF ? 2 : 1) * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap: blocks that share an XCD (same blockIdx % 8) get
  // consecutive tiles, so the column tiles of one row panel hit the same L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  // ROWA kernels are persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of the whole grid
  const int nwg = ROWA ? ((p.M + BM - 1) / BM) * tiles_n : (int)gridDim.x;
  int m0, n0;
  auto tile_origin = [&](int orig) __attribute__((always_inline)) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tile_n = wgid % tiles_n;
    const int tile_m = wgid / tiles_n;
    m0 = tile_m * BM;
    n0 = tile_n * BN;
  };
  int vblock = blockIdx.x;
  DBG_T(0);
  tile_origin(vblock);

  // batch index = outer * batch_inner + inner, each level with its own strides (heads inside images)
  const int zin = p.batch_inner > 1 ? p.batch_inner : 1;
  const int z_o = blockIdx.z / zin, z_i = blockIdx.z - z_o * zin;
  const bf16_t* __restrict__ A = (const bf16_t*)p.a + (int64_t)z_o * p.sa + (int64_t)z_i * p.sa2;
  const bf16_t* __restrict__ W = (const bf16_t*)p.w + (int64_t)z_o * p.sw + (int64_t)z_i * p.sw2;
  const int64_t zo = (int64_t)z_o * p.so + (int64_t)z_i * p.so2, zr = (int64_t)z_o * p.sr;

  // ---- staging state ---------------------------------------------------------------------
  // Both operands are fetched with buffer_load_dwordx4 ... lds through a buffer descriptor: the
  // per-lane address is one 32-bit byte offset, and an offset at or past the descriptor's size
  // reads zeros, which is how padding taps, rows past M / N and the K tail are produced (no
  // branches, no 64-bit address arithmetic in the K loop).
  const int slot = t & 7;          // 16-byte slot inside a 128-byte LDS row
  const int rsub = (PP ? (t & 255) : t) >> 3;   // row inside an RPP-row group
  const int swave = PP ? (wave & 3) : wave;     // staging wave: rows swave * 8 ... + 7 of every RPP-row group
  const int sgrp = PP ? (wave >> 2) : 0;        // PP: waves 0-3 stage activation rows BM / 2 ..., waves 4-7 rows 0 ...
  const int kk = (slot ^ ((rsub >> 1) & 7)) * 8;   // swizzled k offset inside the 64-wide tile
  const int ohw = p.OH * p.OW;
  const PixelIndex pix(p.OH, p.OW);
  const int IHe = p.up2 ? 2 * p.IH : p.IH;
  const int IWe = p.up2 ? 2 * p.IW : p.IW;
  const int ntaps = p.KH * p.KW;
  // Cin % 64 == 0: all lanes of a K-tile sit on the same filter tap, so (tap, channel base) is scalar
  constexpr bool uni = FAST;
  constexpr unsigned OOB = 0x80000000u;   // descriptors never cover more than 2^31 - 1 bytes

  const int w_bytes = p.N * p.Kp * 2;
  const bf16_t* a_base;   // descriptor of the A window of the current tile
  int a_bytes;
  // a_vo: byte offset of (row, tap 0) from the tile's descriptor base (wraps for rows whose tap 0 is
  // padding; only sums that land on valid taps are used).  a_mask: bit tap = tap inside the image;
  // bits 16 / 17 = parity of the tap-0 row / column in the x2-upsampled frame.
  unsigned a_vo[AI], a_mask[AI], w_vo[BI];
  int ch, tap;            // general path: (tap, channel) of this thread's 8-element chunk
  int chb_s = 0, tap_s = 0;   // uniform path: (tap, channel base) of the K-tile
  // staging state of the tile whose origin is (m0, n0)
  auto tile_setup = [&]() __attribute__((always_inline)) {
    const int rs = rsub, kk_ = kk;
    // descriptor base: first input row any tap of this tile's first output row can touch
    int b0, oy0, ox0_;
    pix(m0, b0, oy0, ox0_);
    int by0 = oy0 * p.stride - p.pad_t;
    by0 = by0 < 0 ? 0 : by0;
    if (p.up2) by0 >>= 1;
    const int64_t base_pix = (int64_t)b0 * p.IH * p.IW + (int64_t)by0 * p.IW;
    int64_t rem = ((int64_t)pix.image(p.M) * p.IH * p.IW - base_pix) * p.a_cs * 2;
    rem = rem > 0x7fffffff ? 0x7fffffff : rem;
    a_base = A + base_pix * p.a_cs;
    a_bytes = (int)rem;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = m0 + rs + RPP * i + (PP ? (1 - sgrp) * (BM / 2) : 0);
      unsigned mask = 0, vo = 0;
      if (m < p.M) {
        int b, oy, ox;
        pix(m, b, oy, ox);
        const int by = oy * p.stride - p.pad_t, bx = ox * p.stride - p.pad_l;
        unsigned mx = 0, my = 0;   // taps inside the image along x / y; mask = their outer product
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k < p.KW && (unsigned)(bx + k) < (unsigned)IWe) mx |= 1u << k;
          if (k < p.KH && (unsigned)(by + k) < (unsigned)IHe) my |= 1u << k;
        }
        mask = ((my & 1) ? mx : 0) | ((my & 2) ? mx << p.KW : 0) | ((my & 4) ? mx << (2 * p.KW) : 0);
        int ry = by, rx = bx;
        if (p.up2) {
          mask |= (unsigned)(by & 1) << 16 | (unsigned)(bx & 1) << 17;
          ry = by >> 1;
          rx = bx >> 1;
        }
        const int pix = (b - b0) * p.IH * p.IW + (ry - by0) * p.IW + rx;
        vo = (unsigned)(pix * p.a_cs + p.a_coff + (uni ? kk_ : 0)) * 2u;
        if constexpr (PP) {   // no mask register: 4 edge bits ride in the (zero) low bits of the 16-byte aligned offset
          if (p.KW >= 2) vo |= (oy == 0 ? 1u : 0u) | (oy + 1 == p.OH ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox + 1 == p.OW ? 8u : 0u);
        }
      } else if constexpr (PP) vo = OOB;
      a_vo[i] = vo;
      a_mask[i] = mask;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + rs + RPP * i;
      w_vo[i] = (unsigned)(n * p.Kp + kk_) * 2u;   // rows past N are past the end of the W descriptor
    }
    ch = kk_;
    tap = 0;
    while (ch >= p.Cin) { ch -= p.Cin; ++tap; }
    chb_s = 0;
    tap_s = 0;
  };
  tile_setup();

  // ---- ROWA state --------------------------------------------------------------------------
  // LDS row r of a slot holds flattened input pixel m0 + (kh - 1) * IW - 1 + r, so tap kw of output row j is LDS
  // row j + kw.  The 16-byte chunk q of row r sits in slot q ^ g[(r >> 1) & 7]; this g keeps ds_read_b128 free of
  // bank conflicts for fragment rows shifted by 0, 1 and 2 (the plain (r >> 1) & 7 swizzle only for shift 0).
  constexpr unsigned SWZ = 0x62654210u;
  // edge: 4 bits per row group i of this lane's output pixel: 1 = first image row, 2 = last image row, 4 = first
  // column, 8 = last column.  Tap (kh, kw) is outside the image iff the pixel has an edge bit of
  // (kh == 0 ? 1 : kh == 2 ? 2 : 0) | (kw == 0 ? 4 : kw == 2 ? 8 : 0).
  unsigned edge = 0;
  unsigned ra_vo0 = 0, ra_step = 0;   // byte offset of this thread's first staged pixel, and of RPP pixels
  unsigned wave_edge = 0;             // OR of the edge bits over the wave's lanes and row groups (wave-uniform)
  unsigned gpack = 0;                 // swizzle g of this lane's fragment rows shifted by kw = 0, 1, 2 (4 bits each)
  auto rowa_setup = [&]() __attribute__((always_inline)) {
    const int gsw = (int)(SWZ >> (((rsub >> 1) & 7) * 4)) & 7;
    const int kkA = (slot ^ gsw) * 8;
    const int64_t first = (int64_t)m0 - p.IW - 1;
    const int64_t base_pix = first < 0 ? 0 : first;
    int64_t rem = ((int64_t)pix.image(p.M) * p.IH * p.IW - base_pix) * p.a_cs * 2;
    rem = rem > 0x7fffffff ? 0x7fffffff : rem;
    a_base = A + base_pix * p.a_cs;
    a_bytes = (int)rem;
    const int r0 = (int)(first - base_pix) + rsub;   // may be negative: the byte offset then wraps past the descriptor (zeros)
    {
      const int pr = (lane & 15) >> 1, odd = lane & 1;
      gpack = ((SWZ >> (pr * 4)) & 7) | (((SWZ >> (((pr + odd) & 7) * 4)) & 7) << 4) | (((SWZ >> (((pr + 1) & 7) * 4)) & 7) << 8);
    }
    ra_vo0 = (unsigned)(r0 * p.a_cs * 2 + (p.a_coff + kkA) * 2);
    ra_step = (unsigned)(RPP * p.a_cs * 2);
    unsigned any = 0;
    edge = 0;
    wave_edge = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WM + i * 16 + (lane & 15);
      unsigned e = 0;     // rows past M are never stored
      if (m < p.M) {
        int b, oy, ox;
        pix(m, b, oy, ox);
        e = (oy == 0 ? 1u : 0u) | (oy + 1 == p.OH ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox + 1 == p.OW ? 8u : 0u);
      }
      edge |= e << (4 * i);
      any |= e;
    }
#pragma unroll
    for (int bt = 0; bt < 4; ++bt)
      if (__ballot((any >> bt) & 1) != 0ull) wave_edge |= 1u << bt;
    wave_edge = __builtin_amdgcn_readfirstlane(wave_edge);
  };
  if constexpr (ROWA) rowa_setup();

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const int rd_swz = lr >> 1;   // (row >> 1) & 7 for every fragment row of this lane
  const int rdA = (wm * WM + lr) * 128;
  const int rdB = BM * 128 + (wn * WN + lr) * 128;

  // issue the LDS-DMA loads of K-tile kt into stage buffer `buf`, then advance (tap, channel)
  auto stage = [&](int kt, char* buf) __attribute__((always_inline)) {
    if constexpr (FAST) {
      int kh = 0, kw = 0;
      if (p.KW == 3) { kh = (tap_s * 11) >> 5; kw = tap_s - kh * 3; }
      else if (p.KW == 2) { kh = tap_s >> 1; kw = tap_s & 1; }
      const unsigned bit = 1u << tap_s;
      const unsigned soff = (unsigned)((p.up2 ? 0 : (kh * p.IW + kw) * p.a_cs) + chb_s) * 2u;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        unsigned vo = a_vo[i] + soff;
        if (p.up2) {   // source pixel of the tap in the x2-upsampled frame: offsets depend on the row / column parity
          const int dy = (int)(((a_mask[i] >> 16) & 1) + kh) >> 1, dx = (int)(((a_mask[i] >> 17) & 1) + kw) >> 1;
          vo += (unsigned)((dy * p.IW + dx) * p.a_cs) * 2u;
        }
        vo = (a_mask[i] & bit) ? vo : OOB;
        buffer_lds16(a_base, a_bytes, vo, 0, buf + (wave * 8 + RPP * i) * 128);
      }
      // next (tap, channel base); written as selects so both stay in scalar registers
      const bool tap_major = p.k_order != 0;
      const int t1 = tap_s + 1, c1 = chb_s + 64;
      const bool wrap = tap_major ? (t1 == ntaps) : (c1 >= p.Cin);
      const int nt = tap_major ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
      const int nc = tap_major ? (wrap ? c1 : chb_s) : (wrap ? 0 : c1);
      tap_s = nt;
      chb_s = nc;
    } else {
      int kh = 0, kw = 0;
      if (p.KW == 3) { kh = (tap * 11) >> 5; kw = tap - kh * 3; }
      else if (p.KW == 2) { kh = tap >> 1; kw = tap & 1; }
      const bool tap_ok = tap < ntaps && ch < p.Cin;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        int dy = kh, dx = kw;
        if (p.up2) {
          dy = (int)(((a_mask[i] >> 16) & 1) + kh) >> 1;
          dx = (int)(((a_mask[i] >> 17) & 1) + kw) >> 1;
        }
        unsigned vo = a_vo[i] + (unsigned)((dy * p.IW + dx) * p.a_cs + ch) * 2u;
        vo = (tap_ok && ((a_mask[i] >> tap) & 1)) ? vo : OOB;
        buffer_lds16(a_base, a_bytes, vo, 0, buf + (wave * 8 + RPP * i) * 128);
      }
      if (p.k_order) {
        if (++tap == ntaps) { tap = 0; ch += 64; }
      } else {
        ch += 64;
        while (ch >= p.Cin) { ch -= p.Cin; ++tap; }
      }
    }
    const bool k_ok = (kt << 6) + kk < p.K;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      // a wave stages 8 consecutive rows; waves whose rows lie past the tile (wave-uniform) skip the load
      if ((BN % RPP == 0) || wave * 8 + RPP * i < BN)
        buffer_lds16(W, w_bytes, k_ok ? w_vo[i] : OOB, kt << 7, buf + BM * 128 + (wave * 8 + RPP * i) * 128);
    }
  };

  // PP: group g (0: waves 0-3, 1: waves 4-7) stages the activation rows of the OTHER group and every second 32-row
  // piece of the weight tile.  Tap (kh, kw) of a pixel is outside the image iff the pixel has one of the edge bits
  // sel(kh, kw); the weight rows are an arithmetic progression, so one offset register serves all of them.
  auto stage_pp = [&](const int g, int kt, char* abuf, char* wbuf) __attribute__((always_inline)) {
    int kh = 0, kw = 0;
    unsigned sel = 0;
    if (p.KW == 3) {
      kh = (tap_s * 11) >> 5;
      kw = tap_s - kh * 3;
      sel = (kh == 0 ? 1u : kh == 2 ? 2u : 0u) | (kw == 0 ? 4u : kw == 2 ? 8u : 0u);
    } else if (p.KW == 2) {   // rows y - pad_t, y - pad_t + 1: the first is outside on the first image row (pad_t = 1), the second on the last (pad_t = 0)
      kh = tap_s >> 1;
      kw = tap_s & 1;
      sel = (kh == 0 ? (p.pad_t ? 1u : 0u) : (p.pad_t ? 0u : 2u)) | (kw == 0 ? (p.pad_l ? 4u : 0u) : (p.pad_l ? 0u : 8u));
    }
    const unsigned soff = (unsigned)((kh * p.IW + kw) * p.a_cs + chb_s) * 2u;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const unsigned vo = (a_vo[i] & sel) ? OOB : (a_vo[i] & ~15u) + soff;
      buffer_lds16(a_base, a_bytes, vo, 0, abuf + ((1 - g) * (BM / 2) + swave * 8 + RPP * i) * 128);
    }
    const bool tap_major = p.k_order != 0;
    const int t1 = tap_s + 1, c1 = chb_s + 64;
    const bool wrap = tap_major ? (t1 == ntaps) : (c1 >= p.Cin);
    const int nt = tap_major ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
    const int nc = tap_major ? (wrap ? c1 : chb_s) : (wrap ? 0 : c1);
    tap_s = nt;
    chb_s = nc;
    const unsigned wstep = (unsigned)(RPP * p.Kp) * 2u;
#pragma unroll
    for (int i = g; i < BI; i += 2)
      buffer_lds16(W, w_bytes, w_vo[0] + wstep * (unsigned)i, kt << 7, wbuf + (swave * 8 + RPP * i) * 128);
  };

  // MFMA over one staged 64-deep tile, software-pipelined inside the wave: a tile is 2*NI steps of
  // MI MFMAs (one B fragment against all A fragments of a 32-deep k-step).  Each step first issues
  // the ds_read of the NEXT step's B fragment and, during the first k-step, one share of the second
  // k-step's A fragments, so LDS latency is covered by the step's own MFMAs; only the first
  // MI + 1 reads after the barrier are exposed.
  // pa / pb: this lane's first fragment row of the activation / weight tile; offA0 / offA1: swizzled byte offsets
  // of the two 32-deep k-steps inside an activation row.  mbit != 0 (ROWA): zero the activation fragments of
  // pixels that have one of the edge bits `mbit`, once per fragment, right before its first MFMA.
  auto compute_at = [&](const char* pa, const char* pb, int offA0, int offA1, unsigned mbit) __attribute__((always_inline)) {
    const int off0 = (lq ^ rd_swz) << 4, off1 = ((4 + lq) ^ rd_swz) << 4;
    bf16x8 fa[2][MI], fb[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = *(const bf16x8*)(pa + i * 16 * 128 + offA0);
    fb[0] = *(const bf16x8*)(pb + off0);
#pragma unroll
    for (int st = 0; st < 2 * NI; ++st) {
      const int ks = st / NI, j = st % NI;
      if (st + 1 < 2 * NI) {
        const int ks2 = (st + 1) / NI, j2 = (st + 1) % NI;
        fb[(st + 1) & 1] = *(const bf16x8*)(pb + j2 * 16 * 128 + (ks2 ? off1 : off0));
      }
      if (ks == 0) {
#pragma unroll
        for (int i = j; i < MI; i += NI) fa[1][i] = *(const bf16x8*)(pa + i * 16 * 128 + offA1);
      }
      if constexpr (ROWA) {
        if (j == 0 && mbit) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const unsigned keep = ((edge >> (4 * i)) & mbit) ? 0u : 0xffffffffu;
            u32x4 v = __builtin_bit_cast(u32x4, fa[ks][i]);
            v &= keep;
            fa[ks][i] = __builtin_bit_cast(bf16x8, v);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[st & 1], fa[ks][i], acc[i][j], 0, 0, 0);
    }
  };
  auto compute = [&](const char* buf) __attribute__((always_inline)) {
    compute_at(buf + rdA, buf + rdB, (lq ^ rd_swz) << 4, ((4 + lq) ^ rd_swz) << 4, 0u);
  };

  const int nkt = (p.K + 63) >> 6;
  if constexpr (ROWA) {
    // Macro-step g = (channel chunk, filter row kh): its activation slot serves the three K-steps kw = 0, 1, 2.
    // Weights: two stage buffers, tile of step s + 1 loaded under the MFMAs of step s.  Activations: two slots, the
    // pieces of macro-step g + 1 spread over the three steps of macro-step g (its slot was last read in macro-step
    // g - 1).  One barrier per K-step, as in the double-buffered loop.
    char* acur = smem;
    char* anext = smem + ABYTES;
    char* wcur = smem + 2 * ABYTES;
    char* wnext = wcur + BN * 128;
    auto stage_w = [&](int tp, int chb, char* buf) __attribute__((always_inline)) {
      // column of (tap, chunk) in the packed weight rows: tap-major (k_order 0) or chunk-major (k_order 1)
      const unsigned soff = (unsigned)(p.k_order ? (chb * 9 + tp * 64) : (tp * p.Cin + chb)) * 2u;
#pragma unroll
      for (int i = 0; i < BI; ++i)
        if ((BN % RPP == 0) || wave * 8 + RPP * i < BN)
          buffer_lds16(W, w_bytes, w_vo[i], soff, buf + (wave * 8 + RPP * i) * 128);
    };
    auto stage_a = [&](int piece, int kh, int chb, char* buf) __attribute__((always_inline)) {   // piece AI = the 8 extra rows
      const unsigned soff = (unsigned)(kh * p.IW * p.a_cs + chb) * 2u;
      if (piece < AI || wave == 0)   // the extra piece starts at row BM = RPP * AI
        buffer_lds16(a_base, a_bytes, ra_vo0 + ra_step * (unsigned)piece + soff, 0, buf + (piece < AI ? (wave * 8 + RPP * piece) * 128 : BM * 128));
    };
    const int ng = 3 * (p.Cin >> 6);
    const int rowB = (wn * WN + lr) * 128;
    auto first_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int pc = 0; pc <= AI; ++pc) stage_a(pc, 0, 0, acur);
      stage_w(0, 0, wcur);
    };
    first_stage();
    while (true) {
    int kh = 0, chb = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int g = 0; g < ng; ++g) {
      // chunk-major walk (all three filter rows of a channel chunk, then the next chunk): with chunk-major packed
      // weights the products are summed in the same order as in the per-tap kernels, so every tile shape returns
      // the same bits for a layer and the automatic tile choice cannot make a result depend on the batch size
      int nkh = kh + 1, nchb = chb;
      if (nkh == 3) { nkh = 0; nchb += 64; }
      const bool has_next = g + 1 < ng;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // diagnostic tile 118: shader-clock stamps 0 ... 4 around the parts of one steady-state K-step of the first tile
        const bool dbg = STAMP && g == 2 && kw == 0 && vblock == (int)blockIdx.x;
        if (dbg) DBG_C(0);
        if (kw < 2) stage_w(kh * 3 + kw + 1, chb, wnext);
        else if (has_next) stage_w(nkh * 3, nchb, wnext);
        if (has_next) {
#pragma unroll
          for (int pc = kw; pc <= AI; pc += 3) stage_a(pc, nkh, nchb, anext);
        }
        if (dbg) DBG_C(1);
        // read offsets are recomputed every step from one register: hoisted out of the loop, the per-tap addresses
        // of the three unrolled steps are spilled, and their scratch reloads wait on the LDS-DMA counter
        unsigned gp = gpack;
        asm volatile("" : "+v"(gp));
        const int gsw = (int)(gp >> (4 * kw)) & 7;
        const unsigned sel = (kh == 0 ? 1u : kh == 2 ? 2u : 0u) | (kw == 0 ? 4u : kw == 2 ? 8u : 0u);
        compute_at(acur + rdA + kw * 128, wcur + rowB, (lq ^ gsw) << 4, ((4 + lq) ^ gsw) << 4, sel & wave_edge);
        if (dbg) DBG_C(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg) DBG_C(3);
        __syncthreads();
        if (dbg) DBG_C(4);
        char* tw = wcur; wcur = wnext; wnext = tw;
      }
      char* ta = acur; acur = anext; anext = ta;
      kh = nkh;
      chb = nchb;
    }
    // Next tile of this workgroup: its staging state is set up and its first stage put in flight (into the two
    // slots the last K-step did not read; every wave has passed that step's barrier) BEFORE this tile's epilogue,
    // so the loads land under the output stores and no workgroup hand-over sits between two tiles.
    const int em = m0 + wm * WM, en = n0 + wn * WN;
    vblock += (int)gridDim.x;
    const bool more = vblock < nwg;
    if (more) {
      tile_origin(vblock);
      tile_setup();
      rowa_setup();
      first_stage();
    }
    igemm_epilogue<MI, NI, GNS>(p, acc, em, en, lane, zo, zr);
    if (!more) break;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  } else if constexpr (PP) {
    static_assert(NW == 8 && FAST && DBUF, "ping-pong schedule: 8 waves, uniform taps, two stage buffers");
    // Phase p ends with a barrier.  Group 0 reads K-tile k in phase 2k and multiplies it in phase 2k + 1; group 1 reads
    // it in phase 2k + 1 and multiplies in phase 2k + 2.  In its read phase of tile k, group 0 issues its share of tile
    // k + 1 and group 1 its share of tile k + 2; each waits for its loads at the end of its next multiply phase, which
    // is at least one full phase before anyone reads them.  Both groups pass the same number (2 nkt) of barriers.
    // Barriers inside the loop are bare s_barrier instructions: __syncthreads() carries a fence that drains the
    // vector-memory counter, which would make the staging waves wait for their LDS-DMA loads in the phase that issues
    // them.  The explicit counter waits before each barrier order exactly what the next phase depends on.
    const int grp = wave >> 2;
    bf16x8 fa[2][MI], fb[2][NI];
    char* const wbase = smem + 2 * BM * 128;
    auto abuf = [&](int k) __attribute__((always_inline)) { return smem + (k & 1) * (BM * 128); };
    auto read_tile = [&](const char* ab, const char* wb) __attribute__((always_inline)) {
      const char* pa = ab + rdA;
      const char* pb = wb + (wn * WN + lr) * 128;
      const int off0 = (lq ^ rd_swz) << 4, off1 = ((4 + lq) ^ rd_swz) << 4;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fa[0][i] = *(const bf16x8*)(pa + i * 16 * 128 + off0);
        fa[1][i] = *(const bf16x8*)(pa + i * 16 * 128 + off1);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        fb[0][j] = *(const bf16x8*)(pb + j * 16 * 128 + off0);
        fb[1][j] = *(const bf16x8*)(pb + j * 16 * 128 + off1);
      }
    };
    auto mfma_tile = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // w0 / w1 / w2: weight buffers of K-tiles kt, kt + 1, kt + 2 (kt mod 3 rotation)
    char* w0 = wbase;
    char* w1 = wbase + BN * 128;
    char* w2 = wbase + 2 * BN * 128;
    if (grp == 0) {
      stage_pp(0, 0, abuf(0), w0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      DBG_T(2);
      DBG_C(6);
      for (int kt = 0; kt < nkt; ++kt) {
        read_tile(abuf(kt), w0);
        if (kt + 1 < nkt) stage_pp(0, kt + 1, abuf(kt + 1), w1);   // rows of group 1 last read in phase 2 kt - 1, weight buffer in 2 kt - 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* tw = w0; w0 = w1; w1 = w2; w2 = tw;
      }
    } else {
      stage_pp(1, 0, abuf(0), w0);
      if (nkt > 1) stage_pp(1, 1, abuf(1), w1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      __builtin_amdgcn_s_barrier();
      for (int kt = 0; kt < nkt; ++kt) {
        read_tile(abuf(kt), w0);
        if (kt + 2 < nkt) stage_pp(1, kt + 2, abuf(kt), w2);        // rows of group 0 last read in phase 2 kt, weight buffer in 2 kt - 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kt + 1 < nkt) __builtin_amdgcn_s_barrier();
        char* tw = w0; w0 = w1; w1 = w2; w2 = tw;
      }
    }
  } else if constexpr (DBUF) {
    // two stage buffers, ONE barrier per K-tile: the loads of tile kt+1 are issued before the MFMAs
    // of tile kt and land underneath them.  Buffer (kt+1)&1 was last read by tile kt-1, which every
    // wave has finished when it passed the barrier that ended iteration kt-1.
    DBG_T(1);
    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DBG_T(2);
    DBG_C(6);
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) stage(kt + 1, smem + ((kt + 1) & 1) * STAGE);
      compute(smem + (kt & 1) * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      stage(kt, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(smem);
      __syncthreads();
    }
  }

  DBG_C(7);
  DBG_T(3);
  if constexpr (SMX != 0) {
    igemm_epilogue_smx<MI, NI, SMX>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, zo, (int)blockIdx.z, (n0 / BN) * WAVES_N + wn,
                                    tiles_n * WAVES_N);
    return;
  }
  igemm_epilogue<MI, NI, GNS>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, zo, zr);
  if (STAMP) { DBG_T(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DBG_T(5); }
}

// ---- host side ------------------------------------------------------------------------------
namespace {
struct TileCfg { int bm, bn; };
const TileCfg kTiles[] = {{128, 128}, {128, 64}, {256, 16}, {64, 64}, {256, 64}, {256, 256}, {256, 128},
                          {256, 256},    // 7: 256x256 with 16 waves (64x64 per wave)
                          {512, 128},    // 8: 16 waves, for N = 128 layers
                          {256, 128},    // 9: 16 waves (64x32 per wave)
                          {256, 128},    // 10: 8 waves, single-buffered, two workgroups per CU
                          {256, 160},    // 11: 16 waves (32x80 per wave): the UNet widths 320 ... 2560 are multiples of 160
                          {128, 64},     // 12: tile 1 double-buffered (small grids: few workgroups per CU to overlap with)
                          {64, 64},      // 13: tile 3 double-buffered
                          {128, 128},    // 14: tile 0 double-buffered
                          {384, 128},    // 15: 12 waves (64x64 per wave), shared activation rows (ROWA): 3x3 stride-1 convolutions
                          {192, 128},    // 16: 8 waves (48x64 per wave), double-buffered, two workgroups per CU (2 x 80 KB of LDS)
                          {256, 256},    // 17: 8 waves (64x128 per wave), ping-pong schedule
                          {256, 160},    // 18: 8 waves (64x80 per wave), ping-pong schedule
                          {256, 128}};   // 19: 8 waves (64x64 per wave), ping-pong schedule
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

inline bool fast_path(const lbmk_igemm_desc& d) { return d.Cin % 64 == 0; }
// LBMK_NO_ROWA=1 in the environment keeps the automatic choice on the per-tap tiles (same-box A/B runs)
inline bool rowa_enabled() {
  static const bool on = getenv("LBMK_NO_ROWA") == nullptr;
  return on;
}
// LBMK_NO_PERSIST=1: one workgroup per tile for the ROWA kernels as well (same-box A/B runs)
inline bool persist_enabled() {
  static const bool on = getenv("LBMK_NO_PERSIST") == nullptr;
  return on;
}
// shapes the shared-activation-row (ROWA) kernels take: 3x3, stride 1, one pixel of padding, same-size output
inline bool rowa_ok(const lbmk_igemm_desc& d) {
  return d.KH == 3 && d.KW == 3 && d.stride == 1 && !d.up2 && d.pad_t == 1 && d.pad_l == 1 && d.OH == d.IH && d.OW == d.IW &&
         d.Cin % 64 == 0 && d.batch == 1;
}

// LBMK_NO_PP=1 keeps the automatic choice off the ping-pong tiles (same-box A/B runs)
inline bool pp_enabled() {
  static const bool on = getenv("LBMK_NO_PP") == nullptr;
  return on;
}
// shapes the ping-pong kernels take: their staging keeps 4 edge bits per pixel instead of a 9-tap mask
inline bool pp_ok(const lbmk_igemm_desc& d) {
  if (d.Cin % 64) return false;
  if (d.KH == 1) return d.pad_t == 0 && d.pad_l == 0;
  if (d.KH == 2) return d.stride == 1 && !d.up2 && (d.pad_t | 1) == 1 && (d.pad_l | 1) == 1 && d.OH == d.IH && d.OW == d.IW;
  return d.stride == 1 && !d.up2 && d.pad_t == 1 && d.pad_l == 1 && d.OH == d.IH && d.OW == d.IW;
}

template <int BM, int BN, int WMs, int WNs, bool DBUF, bool STAMP = false, int OCC = 1, bool ROWA = false, bool PP = false>
void launch(const lbmk_igemm_desc& d, hipStream_t s) {
  const int tm = (d.M + BM - 1) / BM, tn = (d.N + BN - 1) / BN;
  int nblocks = tm * tn;
  if (ROWA && persist_enabled() && nblocks > lbmk_num_cus()) nblocks = lbmk_num_cus();   // one persistent workgroup per CU
  dim3 grid((unsigned)nblocks, 1, (unsigned)d.batch);
  if (d.gn_partial) {   // fused GroupNorm statistics: fast path only (checked by the caller)
    if constexpr (BM / WMs == 64 && (BN / WNs == 64 || PP) && !STAMP)
      hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, DBUF, true, false, OCC, true, ROWA, PP>), grid, dim3(64 * WMs * WNs), 0, s, d);
  } else if (fast_path(d))
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, DBUF, true, STAMP, OCC, false, ROWA, PP>), grid, dim3(64 * WMs * WNs), 0, s, d);
  else if constexpr (!ROWA && !PP)
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, DBUF, false, STAMP, OCC>), grid, dim3(64 * WMs * WNs), 0, s, d);
}
}  // namespace

#ifdef LBMK_DIAG
extern "C" int lbmk_debug_igemm_times(unsigned long long* host, int nblocks) {
  if (!host || nblocks <= 0 || nblocks > 65536) return (int)hipErrorInvalidValue;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(igemm_dbg), (size_t)nblocks * 8 * 8);
}
#endif

extern "C" int64_t lbmk_igemm_gn_partial_floats(int M, int N, int Cin, int OH, int OW, int G) {
  if (M <= 0 || N <= 0 || OH <= 0 || OW <= 0 || G <= 0 || N % G || Cin % 64) return -1;
  const int cpg = N / G, ohw = OH * OW;
  if ((cpg != 4 && cpg != 8 && cpg != 16) || ohw % 64 || M % ohw) return -1;
  return (int64_t)(M / 64) * G * 2;
}

extern "C" int lbmk_igemm_smx_parts(int N) { return N > 0 ? ((N + 255) / 256) * 4 : -1; }

extern "C" int lbmk_igemm_tiles(int* bm, int* bn, int cap) {
  for (int i = 0; i < kNumTiles && i < cap; ++i) { bm[i] = kTiles[i].bm; bn[i] = kTiles[i].bn; }
  return kNumTiles;
}

int lbmk_igemm_f32_launch(const lbmk_igemm_desc* d, void* stream);     // fp32.hip

extern "C" int lbmk_igemm(const lbmk_igemm_desc* dp, void* stream) {
  LBMK_REQUIRE(dp != nullptr, "igemm: null descriptor");
  if (dp->in_f32) return lbmk_igemm_f32_launch(dp, stream);      // fp32 mode: f32 operands on the f32-input MFMA
  lbmk_igemm_desc d = *dp;
  LBMK_REQUIRE(d.a && d.w && d.out, "igemm: null operand");
  LBMK_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "igemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  LBMK_REQUIRE(d.batch >= 1 && d.batch <= 65535, "igemm: batch %d", d.batch);
  LBMK_REQUIRE((d.KH == 1 && d.KW == 1) || (d.KH == 2 && d.KW == 2) || (d.KH == 3 && d.KW == 3), "igemm: kernel %dx%d", d.KH, d.KW);
  LBMK_REQUIRE(d.out_up2 >= 0 && d.out_up2 <= 4, "igemm: out_up2 %d", d.out_up2);
  if (d.out_up2) LBMK_REQUIRE(!d.residual && !d.out_f32 && d.batch == 1 && d.act != LBMK_ACT_GEGLU && (int64_t)d.M * 4 < (1LL << 31),
                              "igemm: out_up2 takes an unbatched bf16 launch without residual");
  LBMK_REQUIRE(d.gn_nslab >= 0 && d.gn_strip_off >= 0 && (d.gn_nslab == 0 ? d.gn_strip_off == 0 : d.gn_strip_off + d.OH * d.OW / 64 <= d.gn_nslab),
               "igemm: statistics strips %d + %d of %d", d.gn_strip_off, d.OH * d.OW / 64, d.gn_nslab);
  LBMK_REQUIRE(d.Cin > 0 && d.Cin % 8 == 0, "igemm: Cin=%d must be a multiple of 8", d.Cin);
  LBMK_REQUIRE(d.K == d.KH * d.KW * d.Cin, "igemm: K=%d != KH*KW*Cin", d.K);
  LBMK_REQUIRE((int64_t)(d.N + 512) * d.Kp < (1LL << 30), "igemm: weight matrix %d x %d overflows 32-bit byte offsets", d.N, d.Kp);
  LBMK_REQUIRE((int64_t)d.IH * d.IW * d.a_cs * 2 <= (1LL << 30), "igemm: one input image exceeds 1 GiB (32-bit byte offsets)");
  LBMK_REQUIRE(d.IH < 32767 && d.IW < 32767 && d.OH * d.stride < 32767 && d.OW * d.stride < 32767, "igemm: image side >= 32767");
  LBMK_REQUIRE(d.Kp >= d.K && d.Kp % 8 == 0, "igemm: Kp=%d (row stride of w) must be a multiple of 8 >= K", d.Kp);
  LBMK_REQUIRE(d.a_cs % 8 == 0 && d.a_coff % 8 == 0 && d.a_coff >= 0 && d.a_coff + d.Cin <= d.a_cs,
               "igemm: channel window off=%d Cin=%d stride=%d", d.a_coff, d.Cin, d.a_cs);
  LBMK_REQUIRE(d.sa % 8 == 0 && d.sw % 8 == 0, "igemm: batch strides must keep 16-byte alignment");
  LBMK_REQUIRE(lbmk_aligned(d.a, 16) && lbmk_aligned(d.w, 16), "igemm: a / w must be 16-byte aligned");
  LBMK_REQUIRE(d.IH > 0 && d.IW > 0 && d.OH > 0 && d.OW > 0 && d.stride >= 1, "igemm: spatial dims");
  LBMK_REQUIRE(d.M % (d.OH * d.OW) == 0, "igemm: M=%d is not a whole number of %dx%d images", d.M, d.OH, d.OW);
  LBMK_REQUIRE((int64_t)(d.M / (d.OH * d.OW)) * d.IH * d.IW < (1LL << 31), "igemm: input pixel count overflows int32");
  LBMK_REQUIRE(d.act >= 0 && d.act <= 3, "igemm: act %d", d.act);
  LBMK_REQUIRE(d.k_order == 0 || (d.k_order == 1 && d.Cin % 64 == 0), "igemm: k_order %d needs Cin %% 64 == 0", d.k_order);
  LBMK_REQUIRE(d.ldo >= (d.act == LBMK_ACT_GEGLU ? d.N / 2 : d.N), "igemm: ldo=%d too small", d.ldo);
  LBMK_REQUIRE(lbmk_aligned(d.out, d.out_f32 ? 16 : 8), "igemm: out misaligned");
  LBMK_REQUIRE(!d.residual || lbmk_aligned(d.residual, 8), "igemm: residual misaligned");
  LBMK_REQUIRE(!d.rowvec || lbmk_aligned(d.rowvec, 8), "igemm: rowvec misaligned");
  LBMK_REQUIRE(!(d.bias_per_m && d.act == LBMK_ACT_GEGLU), "igemm: bias_per_m with GEGLU");
  if (d.act == LBMK_ACT_GEGLU) {
    LBMK_REQUIRE(d.N % 32 == 0, "igemm: GEGLU needs N %% 32 == 0 (N=%d)", d.N);
    LBMK_REQUIRE(!d.residual && !d.rowvec && !d.out_f32, "igemm: GEGLU epilogue is bias-only, bf16");
  }
  if ((d.so % 4) || (d.sr % 4)) LBMK_REQUIRE(false, "igemm: so/sr must be multiples of 4");
  if (d.batch_inner > 1) {
    LBMK_REQUIRE(d.batch % d.batch_inner == 0 && d.sa2 % 8 == 0 && d.sw2 % 8 == 0 && d.so2 % 4 == 0 && !d.residual,
                 "igemm: inner batch %d of %d / inner strides", d.batch_inner, d.batch);
  }
  const bool gn = d.gn_partial != nullptr;
  if (gn) {
    LBMK_REQUIRE(lbmk_igemm_gn_partial_floats(d.M, d.N, d.Cin, d.OH, d.OW, d.gn_groups) > 0,
                 "igemm: shape M=%d N=%d Cin=%d %dx%d G=%d cannot carry fused GroupNorm statistics", d.M, d.N, d.Cin, d.OH, d.OW, d.gn_groups);
    LBMK_REQUIRE(!d.out_f32 && d.act != LBMK_ACT_GEGLU && d.batch == 1 && d.ldo % 4 == 0 && lbmk_aligned(d.gn_partial, 8),
                 "igemm: fused GroupNorm statistics need a bf16, non-GEGLU, unbatched output");
    LBMK_REQUIRE((!d.residual || (d.ldr % 8 == 0 && lbmk_aligned(d.residual, 16))) && (!d.rowvec || d.rowvec_ld % 4 == 0) &&
                 d.ldo % 8 == 0 && d.N % 8 == 0 && lbmk_aligned(d.out, 16),
                 "igemm: gn statistics need 16-byte aligned output / residual rows");
  }

  if (d.smx_mode) {
    // two-pass row softmax over a scores GEMM: the 16-wave 256x256 tile with the reducing epilogues
    LBMK_REQUIRE(d.smx_mode == 1 || d.smx_mode == 2, "igemm: smx_mode %d", d.smx_mode);
    LBMK_REQUIRE(fast_path(d) && d.KH == 1 && !gn && !d.bias && !d.residual && !d.rowvec && d.act == LBMK_ACT_NONE && !d.out_up2,
                 "igemm: the softmax passes take a plain 1x1 GEMM with K %% 64 == 0");
    LBMK_REQUIRE(d.smx_part && lbmk_aligned(d.smx_part, 4) && (d.smx_mode == 1 || (d.smx_max && !d.out_f32 && d.N % 4 == 0 && d.ldo % 4 == 0)),
                 "igemm: softmax pass %d operands", d.smx_mode);
    const int tm = (d.M + 255) / 256, tn = (d.N + 255) / 256;
    dim3 grid((unsigned)(tm * tn), 1, (unsigned)d.batch);
    if (d.smx_mode == 1)
      hipLaunchKernelGGL((igemm_kernel<256, 256, 4, 4, true, true, false, 1, false, false, false, 1>), grid, dim3(1024), 0, (hipStream_t)stream, d);
    else
      hipLaunchKernelGGL((igemm_kernel<256, 256, 4, 4, true, true, false, 1, false, false, false, 2>), grid, dim3(1024), 0, (hipStream_t)stream, d);
    return lbmk_check_launch("igemm");
  }
  int tile = d.tile;
  if (tile < 0) {
    if (d.act != LBMK_ACT_GEGLU && d.N <= 16) tile = 2;
    else if (rowa_enabled() && rowa_ok(d) && d.N >= 96 && d.N <= 128 && d.M >= 65536) tile = 15;   // measured 6-20 % ahead of every other tile on these
    else {
      // Cost model over the production tiles: time of the busiest CU.  A CU holding w workgroups of a
      // tile runs them `resident` at a time; a round of n co-resident workgroups costs a fixed part
      // (prologue + epilogue) plus, per 64-deep K step, max(n * tc, L [+ tc when single-buffered]):
      // tc = MFMA/LDS time of one workgroup's step, L = exposed load + barrier latency, which only
      // co-resident workgroups (or the second stage buffer) can hide.  Constants (microseconds) are
      // least-squares fits to measured GEMMs, M 1k..16k, N 320..10k, K 640..5k; padded rows / columns
      // are paid in full.
      struct Cand { int tile, bm, bn, resident; bool dbuf; float a, o, tc, L; };
      static const Cand cands[] = {{7, 256, 256, 1, true, 0.f, 12.1f, 1.01f, 1.20f},  {11, 256, 160, 1, true, 0.f, 8.0f, 1.02f, 1.05f},
                                   {8, 512, 128, 1, true, 0.f, 12.1f, 1.14f, 1.20f},  {10, 256, 128, 2, false, 0.f, 8.3f, 0.87f, 0.38f},
                                   {0, 128, 128, 4, false, 2.8f, 4.8f, 0.55f, 0.34f}, {1, 128, 64, 4, false, 5.0f, 1.9f, 0.34f, 0.23f},
                                   {3, 64, 64, 8, false, 2.5f, 2.2f, 0.25f, 0.30f},   {12, 128, 64, 3, true, 1.9f, 4.5f, 0.34f, 0.55f},
                                   {13, 64, 64, 4, true, 3.3f, 2.0f, 0.24f, 0.42f},   {14, 128, 128, 2, true, 0.f, 7.9f, 0.20f, 0.84f},
                                   {16, 192, 128, 2, true, 0.f, 12.0f, 0.58f, 0.90f},    // offered to batched launches only (below)
                                   // ping-pong tiles: about 8 % less time per K step than their 16-wave counterparts (in-kernel
                                   // stamps: 1.29 vs 1.41 us at 256x256), a slightly longer epilogue (8 waves store the tile)
                                   {17, 256, 256, 1, true, 0.f, 12.8f, 1.01f, 1.10f},  {18, 256, 160, 1, true, 0.f, 7.5f, 1.02f, 0.99f},
                                   // shared activation rows, one persistent workgroup per CU (3x3 stride-1 convolutions up to 256 columns)
                                   {15, 384, 128, 1, true, 0.f, 8.1f, 1.48f, 1.48f}};
      const int cus = lbmk_num_cus();
      const int nkt = (d.K + 63) / 64;
      double best = 1e300;
      tile = 0;
      for (const Cand& c : cands) {
        if (gn && !(c.tile == 0 || c.tile == 7 || c.tile == 8 || c.tile == 10 || c.tile == 14 || c.tile == 15 || c.tile == 17)) continue;   // statistics strips are 64 rows per wave, column groups in pairs
        if (d.act == LBMK_ACT_GEGLU && (c.tile == 11 || c.tile == 18)) continue;   // odd number of column groups per wave
        if ((c.tile == 17 || c.tile == 18) && !(pp_enabled() && pp_ok(d))) continue;
        if (c.tile == 15 && !(rowa_enabled() && rowa_ok(d) && d.N <= 256)) continue;   // wider layers: the weight tile dominates its staging, 5-25 % behind
        if (c.tile == 16 && d.batch < 8) continue;   // measured ahead only on the per-image V^T projections (many small problems)
        const long blocks = (long)((d.M + c.bm - 1) / c.bm) * ((d.N + c.bn - 1) / c.bn) * d.batch;
        long w = (blocks + cus - 1) / cus;
        const long full = w / c.resident;
        const int rest = (int)(w - full * c.resident);
        auto round_cost = [&](int n) { const double lat = c.dbuf ? c.L : c.L + c.tc; const double st = n * c.tc > lat ? n * c.tc : lat; return c.o + nkt * st; };
        double cost = c.a + full * round_cost(c.resident) + (rest ? round_cost(rest) : 0.0);
        if (cost < best) { best = cost; tile = c.tile; }
      }
    }
  }
#ifdef LBMK_DIAG
  if (tile == 115) { launch<512, 128, 8, 2, true, true>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 116) { launch<256, 256, 4, 4, true, true>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 117) { launch<256, 256, 4, 2, true, true, 1, false, true>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 118) { launch<384, 128, 6, 2, true, true, 1, true>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
#endif
  LBMK_REQUIRE(tile >= 0 && tile < kNumTiles, "igemm: tile %d", tile);
  if (d.act == LBMK_ACT_GEGLU) LBMK_REQUIRE(tile != 2 && tile != 11 && tile != 18, "igemm: GEGLU needs an even number (>= 2) of 16-column groups per wave");
  if (gn) LBMK_REQUIRE(tile == 0 || tile == 4 || tile == 7 || tile == 8 || tile == 10 || tile == 14 || tile == 15 || tile == 17 || tile == 19, "igemm: tile %d has no 64x64 wave strips", tile);
  if (tile == 15) LBMK_REQUIRE(rowa_ok(d), "igemm: tile %d takes 3x3 stride-1 same-size convolutions with Cin %% 64 == 0 only", tile);
  if (tile >= 17 && tile <= 19) LBMK_REQUIRE(pp_ok(d), "igemm: tile %d (ping-pong schedule) takes 1x1 and 2x2 / 3x3 stride-1 same-size layers with Cin %% 64 == 0 only", tile);
  hipStream_t s = (hipStream_t)stream;
  switch (tile) {
    case 0: launch<128, 128, 2, 2, false, false, 4>(d, s); break;
    case 1: launch<128, 64, 2, 2, false, false, 4>(d, s); break;
    case 2: launch<256, 16, 4, 1, false>(d, s); break;
    case 3: launch<64, 64, 2, 2, false>(d, s); break;
    case 4: launch<256, 64, 4, 1, false, false, 4>(d, s); break;
    case 5: launch<256, 256, 2, 4, true>(d, s); break;
    case 6: launch<256, 128, 4, 2, true>(d, s); break;
    case 7: launch<256, 256, 4, 4, true>(d, s); break;
    case 8: launch<512, 128, 8, 2, true>(d, s); break;
    case 9: launch<256, 128, 4, 4, true>(d, s); break;
    case 10: launch<256, 128, 4, 2, false, false, 2>(d, s); break;
    case 11: launch<256, 160, 8, 2, true>(d, s); break;
    case 12: launch<128, 64, 2, 2, true, false, 3>(d, s); break;
    case 13: launch<64, 64, 2, 2, true, false, 4>(d, s); break;
    case 14: launch<128, 128, 2, 2, true, false, 2>(d, s); break;
    case 15: launch<384, 128, 6, 2, true, false, 1, true>(d, s); break;
    case 16: launch<192, 128, 4, 2, true, false, 2>(d, s); break;
    case 17: launch<256, 256, 4, 2, true, false, 1, false, true>(d, s); break;
    case 18: launch<256, 160, 4, 2, true, false, 1, false, true>(d, s); break;
    case 19: launch<256, 128, 4, 2, true, false, 1, false, true>(d, s); break;
  }
  return lbmk_check_launch("igemm");
}
