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

__device__ __attribute__((aligned(16))) uint32_t igemm_zero_page[8];
__device__ unsigned long long igemm_dbg[65536 * 6];  // TEMP timing instrumentation (ABL == 5)
#define DBG_T(i) do { if (ABL == 5 && threadIdx.x == 0 && blockIdx.x < 65536) igemm_dbg[blockIdx.x * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)


// Epilogue shared by every tile shape: acc[i][j] is the 16x16 tile (i, j) of this wave, whose first
// output row / packed column are mbase / nbase.
// Loads are batched ahead of the stores on purpose: `out` may alias anything as far as the compiler
// knows, so a load written after a store waits for it.  The bias of all NI column groups is read once,
// and the residual / rowvec operands of row group i+1 are in flight while row group i is stored.
template <int MI, int NI>
__device__ __forceinline__ void igemm_epilogue(const lbmk_igemm_desc& p, f32x4 (&acc)[MI][NI], int mbase,
                                               int nbase, int lane, int z) {
  const int lr = lane & 15, lq = lane >> 4;
  const int ohw = p.OH * p.OW;
  const int64_t zo = (int64_t)z * p.so, zr = (int64_t)z * p.sr;
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
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nbase + j * 16 + lq * 4 + e;
        bv[j][e] = (bias && !p.bias_per_m && n < p.N) ? bias[n] : 0.f;
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

  // output column of element 0 of column group j (GEGLU: value / gate groups alternate, j even is kept)
  auto out_col = [&](int j) { return geglu ? ((nbase + j * 16) >> 1) + lq * 4 : nbase + j * 16 + lq * 4; };

  if (vec_ok) {
    const bool has_add = res || rowvec;
    u16x4 rr[NI], rv[NI];
    auto fetch = [&](int i) {
      const int m = mbase + i * 16 + lr;
      const bool row_ok = m < p.M;
      const int bimg = (rowvec && row_ok) ? m / ohw : 0;
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
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (geglu && (j & 1)) continue;
        const int no = out_col(j);
        if (no >= n_out) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][e];
          if (geglu) v[e] *= gelu_erf_f(acc[i][(j + 1) % NI][e]);
          else if (p.act == LBMK_ACT_SILU) v[e] = silu_f(v[e]);
        }
        if (p.out_f32) {
          float* o = (float*)p.out + zo + (int64_t)m * p.ldo + no;
          *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          u16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.v[e] = f2bf(v[e]);
          *(u16x4*)((bf16_t*)p.out + zo + (int64_t)m * p.ldo + no) = ov;
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
    const int bimg = rowvec ? m / ohw : 0;
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
        if (p.out_f32) ((float*)p.out)[zo + (int64_t)m * p.ldo + no + e] = val;
        else ((bf16_t*)p.out)[zo + (int64_t)m * p.ldo + no + e] = f2bf(val);
      }
    }
  }
}

// PERSIST (double-buffered tiles only): the grid is one workgroup per CU slot and each workgroup walks
// tiles  blockIdx.x, blockIdx.x + gridDim.x, ...  The first K-tile of the next output tile is issued
// before the epilogue of the current one, so neither the workgroup turnover of a 16-wave / 160 KB-LDS
// block nor the load latency of its first stage is paid once per output tile.
// FAST: Cin % 64 == 0 -- every lane of a K-tile is on the same filter tap, so the
// tap / channel walk is scalar and a load costs three vector ALU instructions.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DBUF, int ABL = 0, bool PERSIST = false, bool FAST = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_kernel(const lbmk_igemm_desc p) {
  static_assert(!PERSIST || DBUF, "persistent tiles are double-buffered");
  constexpr int NW = WAVES_M * WAVES_N;         // waves per workgroup (4 or 8)
  constexpr int NT = 64 * NW;
  constexpr int RPP = NT / 8;                   // tile rows staged by one LDS-DMA instruction of the block
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int AI = BM / RPP;                  // LDS-DMA loads per thread for the A tile
  constexpr int BI = (BN >= RPP) ? BN / RPP : 1;
  constexpr int STAGE = (BM + BN) * 128;        // bytes of one K-tile (64 deep) of both operands
  static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % RPP == 0, "tile shape");
  // all LDS in ONE array: a second __shared__ object makes hipcc drain vmcnt before every ds_read
  __shared__ __attribute__((aligned(16))) char smem[(DBUF ? 2 : 1) * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap: blocks that share an XCD (same blockIdx % 8) get
  // consecutive tiles, so the column tiles of one row panel hit the same L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  // number of (virtual) workgroups of the remap: every output tile, also when fewer blocks walk them
  const int nwg = PERSIST ? tiles_n * ((p.M + BM - 1) / BM) : (int)gridDim.x;
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

  const int z = blockIdx.z;
  const bf16_t* __restrict__ A = (const bf16_t*)p.a + (int64_t)z * p.sa;
  const bf16_t* __restrict__ W = (const bf16_t*)p.w + (int64_t)z * p.sw;

  // ---- staging state ---------------------------------------------------------------------
  // Both operands are fetched with buffer_load_dwordx4 ... lds through a buffer descriptor: the
  // per-lane address is one 32-bit byte offset, and an offset at or past the descriptor's size
  // reads zeros, which is how padding taps, rows past M / N and the K tail are produced (no
  // branches, no 64-bit address arithmetic in the K loop).
  const int slot = t & 7;          // 16-byte slot inside a 128-byte LDS row
  const int rsub = t >> 3;         // row inside an RPP-row group
  const int kk = (slot ^ ((rsub >> 1) & 7)) * 8;   // swizzled k offset inside the 64-wide tile
  const int ohw = p.OH * p.OW;
  const int IHe = p.up2 ? 2 * p.IH : p.IH;
  const int IWe = p.up2 ? 2 * p.IW : p.IW;
  const bool b_loader = (BN >= RPP) || (t < BN * 8);
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
    int tt = t;
    if (PERSIST) asm volatile("" : "+v"(tt));   // recompute per tile instead of keeping it live across the K loop
    const int rs = PERSIST ? (tt >> 3) : rsub;
    const int kk_ = PERSIST ? (((tt & 7) ^ ((rs >> 1) & 7)) * 8) : kk;
    // descriptor base: first input row any tap of this tile's first output row can touch
    const int b0 = m0 / ohw;
    const int oy0 = (m0 - b0 * ohw) / p.OW;
    int by0 = oy0 * p.stride - p.pad_t;
    by0 = by0 < 0 ? 0 : by0;
    if (p.up2) by0 >>= 1;
    const int64_t base_pix = (int64_t)b0 * p.IH * p.IW + (int64_t)by0 * p.IW;
    int64_t rem = ((int64_t)(p.M / ohw) * p.IH * p.IW - base_pix) * p.a_cs * 2;
    rem = rem > 0x7fffffff ? 0x7fffffff : rem;
    a_base = A + base_pix * p.a_cs;
    a_bytes = (int)rem;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = m0 + rs + RPP * i;
      unsigned mask = 0, vo = 0;
      if (m < p.M) {
        const int b = m / ohw, r2 = m - b * ohw;
        const int oy = r2 / p.OW, ox = r2 - oy * p.OW;
        const int by = oy * p.stride - p.pad_t, bx = ox * p.stride - p.pad_l;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            if (kh < p.KH && kw < p.KW && (unsigned)(by + kh) < (unsigned)IHe && (unsigned)(bx + kw) < (unsigned)IWe)
              mask |= 1u << (kh * p.KW + kw);
        int ry = by, rx = bx;
        if (p.up2) {
          mask |= (unsigned)(by & 1) << 16 | (unsigned)(bx & 1) << 17;
          ry = by >> 1;
          rx = bx >> 1;
        }
        const int pix = (b - b0) * p.IH * p.IW + (ry - by0) * p.IW + rx;
        vo = (unsigned)(pix * p.a_cs + p.a_coff + (uni ? kk_ : 0)) * 2u;
      }
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
    if (b_loader) {
      const bool k_ok = (kt << 6) + kk < p.K;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        buffer_lds16(W, w_bytes, k_ok ? w_vo[i] : OOB, kt << 7, buf + BM * 128 + (wave * 8 + RPP * i) * 128);
      }
    }
  };

  // MFMA over one staged 64-deep tile, software-pipelined inside the wave: a tile is 2*NI steps of
  // MI MFMAs (one B fragment against all A fragments of a 32-deep k-step).  Each step first issues
  // the ds_read of the NEXT step's B fragment and, during the first k-step, one share of the second
  // k-step's A fragments, so LDS latency is covered by the step's own MFMAs; only the first
  // MI + 1 reads after the barrier are exposed.
  auto compute = [&](const char* buf) __attribute__((always_inline)) {
    const int off0 = (lq ^ rd_swz) << 4, off1 = ((4 + lq) ^ rd_swz) << 4;
    const char* pa = buf + rdA;
    const char* pb = buf + rdB;
    if constexpr (PERSIST) {
      // one set of A fragments: the second k-step's fragment i replaces the first's right after its
      // last use (frees 4*MI registers for the tile walk state)
      bf16x8 fa[MI], fb[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(pa + i * 16 * 128 + off0);
      fb[0] = *(const bf16x8*)(pb + off0);
#pragma unroll
      for (int st = 0; st < 2 * NI; ++st) {
        const int ks = st / NI, j = st % NI;
        if (st + 1 < 2 * NI) {
          const int ks2 = (st + 1) / NI, j2 = (st + 1) % NI;
          fb[(st + 1) & 1] = *(const bf16x8*)(pb + j2 * 16 * 128 + (ks2 ? off1 : off0));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[st & 1], fa[i], acc[i][j], 0, 0, 0);
          if (ks == 0 && j == NI - 1) fa[i] = *(const bf16x8*)(pa + i * 16 * 128 + off1);
        }
      }
      return;
    }
    bf16x8 fa[2][MI], fb[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = *(const bf16x8*)(pa + i * 16 * 128 + off0);
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
        for (int i = j; i < MI; i += NI) fa[1][i] = *(const bf16x8*)(pa + i * 16 * 128 + off1);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[st & 1], fa[ks][i], acc[i][j], 0, 0, 0);
    }
  };

  const int nkt = (p.K + 63) >> 6;
  if constexpr (PERSIST) {
    stage(0, smem);
    while (true) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage(kt + 1, smem + ((kt + 1) & 1) * STAGE);
        compute(smem + (kt & 1) * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      // every wave is past the last barrier: both stage buffers are free for the next tile
      const int cm0 = m0, cn0 = n0;
      vblock += gridDim.x;
      const bool more = vblock < nwg;
      if (more) {
        tile_origin(vblock);
        tile_setup();
        stage(0, smem);
      }
      igemm_epilogue<MI, NI>(p, acc, cm0 + wm * WM, cn0 + wn * WN, lane, z);
      if (!more) return;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    for (int kt = 0; kt < nkt; ++kt) {
      // ABL != 0: timing-only ablations (1: no loads in the loop, 2: no MFMA work, 3: no loads and
      // no barrier); results are invalid
      if (ABL != 1 && ABL != 3 && kt + 1 < nkt) stage(kt + 1, smem + ((kt + 1) & 1) * STAGE);
      if (ABL != 2) compute(smem + (kt & 1) * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (ABL != 3) __syncthreads();
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

  DBG_T(3);
  igemm_epilogue<MI, NI>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, z);
  if (ABL == 5) { DBG_T(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DBG_T(5); }
}

// ------------------------------------------------------------------------------------------------
// Deep-prefetch variant: 32-deep K tiles in NSTAGE LDS stages.  NSTAGE-1 tiles of LDS-DMA loads are
// in flight while one is computed; a counted s_waitcnt vmcnt(N) retires only the oldest tile and a raw
// s_barrier (no vmcnt(0) drain) publishes it, so the L2 -> LDS path never idles between tiles.
// LDS rows are 64 bytes (4 slots of 16 B, 4 rows per 256-byte bank row); the slot permutation
// f(g) = [0,2,3,1][(row >> 2) & 3] makes every 16-lane ds_read_b128 group conflict-free.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int swz32(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void igemm_k32_kernel(const lbmk_igemm_desc p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = 64 * NW;
  constexpr int RPP = NT / 4;                   // rows staged per LDS-DMA instruction of the block
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int AI = BM / RPP, BI = BN / RPP;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int LPT = AI + BI;                  // LDS-DMA loads per thread per tile
  static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile shape");
  static_assert((NSTAGE - 2) * LPT <= 15, "vmcnt immediates");
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tiles_n = (p.N + BN - 1) / BN;
  int wgid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tile_n = wgid % tiles_n;
  const int tile_m = wgid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const bf16_t* __restrict__ A = (const bf16_t*)p.a + (int64_t)z * p.sa;
  const bf16_t* __restrict__ W = (const bf16_t*)p.w + (int64_t)z * p.sw;

  const int slot = t & 3, rsub = t >> 2;
  const int kk = (slot ^ swz32(rsub)) * 8;
  const int ohw = p.OH * p.OW;
  const int IHe = p.up2 ? 2 * p.IH : p.IH;
  const int IWe = p.up2 ? 2 * p.IW : p.IW;

  int a_by[AI], a_bx[AI], a_boff[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + rsub + RPP * i;
    if (m < p.M) {
      const int b = m / ohw, rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_by[i] = oy * p.stride - p.pad_t;
      a_bx[i] = ox * p.stride - p.pad_l;
      a_boff[i] = b * p.IH * p.IW;
    } else {
      a_by[i] = -(1 << 28);
      a_bx[i] = 0;
      a_boff[i] = 0;
    }
  }
  const bf16_t* w_ptr[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + rsub + RPP * i;
    w_ptr[i] = n < p.N ? W + (int64_t)n * p.Kp + kk : nullptr;
  }
  int ch = kk, tap = 0;
  while (ch >= p.Cin) { ch -= p.Cin; ++tap; }
  const int ntaps = p.KH * p.KW;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  const int rd_off = (lq ^ swz32(lr)) << 4;
  const int rdA = (wm * WM + lr) * 64 + rd_off;
  const int rdB = BM * 64 + (wn * WN + lr) * 64 + rd_off;

  auto stage = [&](int kt, char* buf) {
    int kh = 0, kw = 0;
    if (p.KW == 3) { kh = (tap * 11) >> 5; kw = tap - kh * 3; }
    const bool tap_ok = tap < ntaps && ch < p.Cin;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int iy = a_by[i] + kh, ix = a_bx[i] + kw;
      const bool ok = tap_ok && (unsigned)iy < (unsigned)IHe && (unsigned)ix < (unsigned)IWe;
      if (p.up2) { iy >>= 1; ix >>= 1; }
      const bf16_t* src = ok
          ? A + ((int64_t)(a_boff[i] + iy * p.IW + ix) * p.a_cs + p.a_coff + ch)
          : (const bf16_t*)igemm_zero_page;
      glds16(src, buf + (wave * 16 + RPP * i) * 64);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const bf16_t* src = (w_ptr[i] && (kt << 5) + kk < p.K) ? w_ptr[i] + (kt << 5)
                                                               : (const bf16_t*)igemm_zero_page;
      glds16(src, buf + BM * 64 + (wave * 16 + RPP * i) * 64);
    }
    if (p.k_order) {          // chunk-major: (64-channel chunk, tap, 32-channel half)
      ch += 32;
      if ((ch & 63) < 32) {   // finished both halves of this (chunk, tap)
        ch -= 64;
        if (++tap == ntaps) { tap = 0; ch += 64; }
      }
    } else {
      ch += 32;
      while (ch >= p.Cin) { ch -= p.Cin; ++tap; }
    }
  };

  auto compute = [&](const char* buf) {
    const char* pa = buf + rdA;
    const char* pb = buf + rdB;
    bf16x8 fa[MI], fb[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *(const bf16x8*)(pa + i * 16 * 64);
    fb[0] = *(const bf16x8*)pb;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (j + 1 < NI) fb[(j + 1) & 1] = *(const bf16x8*)(pb + (j + 1) * 16 * 64);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j & 1], fa[i], acc[i][j], 0, 0, 0);
    }
  };

  const int nkt = (p.K + 31) >> 5;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nkt) stage(s, smem + s * STAGE);
  int rd_stage = 0, wr_stage = NSTAGE - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    // tiles issued after tile kt and still allowed in flight: min(NSTAGE-2, nkt-1-kt)
    const int ahead = nkt - 1 - kt;
    if (ahead >= NSTAGE - 2) wait_vmcnt<(NSTAGE - 2) * LPT>();
    else if (NSTAGE > 3 && ahead == NSTAGE - 3) wait_vmcnt<(NSTAGE > 3 ? NSTAGE - 3 : 0) * LPT>();
    else if (NSTAGE > 4 && ahead == NSTAGE - 4) wait_vmcnt<(NSTAGE > 4 ? NSTAGE - 4 : 0) * LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // stage wr_stage was read by tile kt-1, which every wave finished before this barrier
    if (kt + NSTAGE - 1 < nkt) stage(kt + NSTAGE - 1, smem + wr_stage * STAGE);
    compute(smem + rd_stage * STAGE);
    rd_stage = rd_stage + 1 == NSTAGE ? 0 : rd_stage + 1;
    wr_stage = wr_stage + 1 == NSTAGE ? 0 : wr_stage + 1;
  }
  igemm_epilogue<MI, NI>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, z);
}


// ---- host side ------------------------------------------------------------------------------
namespace {
struct TileCfg { int bm, bn; };
const TileCfg kTiles[] = {{128, 128}, {128, 64}, {256, 16}, {64, 64}, {256, 64}, {256, 256}, {256, 128},
                          {256, 256},    // 7: 256x256 with 16 waves (64x64 per wave)
                          {512, 128},    // 8: 16 waves, for N = 128 layers
                          {256, 128},    // 9: 16 waves (64x32 per wave)
                          {256, 256},    // 10: 16 waves, 32-deep K tiles, 4 stages, counted vmcnt
                          {256, 256},    // 11: same, 5 stages
                          {256, 256},    // 12: tile 7, persistent workgroups
                          {512, 128}};   // 13: tile 8, persistent workgroups
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

inline bool fast_path(const lbmk_igemm_desc& d) { return d.Cin % 64 == 0; }

template <int BM, int BN, int WMs, int WNs, bool DBUF, int ABL = 0>
void launch(const lbmk_igemm_desc& d, hipStream_t s) {
  const int tm = (d.M + BM - 1) / BM, tn = (d.N + BN - 1) / BN;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)d.batch);
  if (fast_path(d))
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, DBUF, ABL, false, true>), grid, dim3(64 * WMs * WNs), 0, s, d);
  else
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, DBUF, ABL, false, false>), grid, dim3(64 * WMs * WNs), 0, s, d);
}
// one workgroup per CU (16 waves, >= 128 KB of LDS each), walking the output tiles
template <int BM, int BN, int WMs, int WNs>
void launch_persist(const lbmk_igemm_desc& d, hipStream_t s) {
  const int tm = (d.M + BM - 1) / BM, tn = (d.N + BN - 1) / BN;
  const int nt = tm * tn;
  dim3 grid((unsigned)(nt < lbmk_num_cus() ? nt : lbmk_num_cus()), 1, (unsigned)d.batch);
  if (fast_path(d))
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, true, 0, true, true>), grid, dim3(64 * WMs * WNs), 0, s, d);
  else
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WMs, WNs, true, 0, true, false>), grid, dim3(64 * WMs * WNs), 0, s, d);
}
template <int BM, int BN, int WMs, int WNs, int NSTAGE>
void launch_k32(const lbmk_igemm_desc& d, hipStream_t s) {
  const int tm = (d.M + BM - 1) / BM, tn = (d.N + BN - 1) / BN;
  dim3 grid((unsigned)(tm * tn), 1, (unsigned)d.batch);
  hipLaunchKernelGGL((igemm_k32_kernel<BM, BN, WMs, WNs, NSTAGE>), grid, dim3(64 * WMs * WNs), 0, s, d);
}
}  // namespace

extern "C" int lbmk_debug_igemm_times(unsigned long long* host, int nblocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(igemm_dbg), (size_t)nblocks * 6 * 8);
}

extern "C" int lbmk_igemm_tiles(int* bm, int* bn, int cap) {
  for (int i = 0; i < kNumTiles && i < cap; ++i) { bm[i] = kTiles[i].bm; bn[i] = kTiles[i].bn; }
  return kNumTiles;
}

extern "C" int lbmk_igemm(const lbmk_igemm_desc* dp, void* stream) {
  LBMK_REQUIRE(dp != nullptr, "igemm: null descriptor");
  lbmk_igemm_desc d = *dp;
  LBMK_REQUIRE(d.a && d.w && d.out, "igemm: null operand");
  LBMK_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "igemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
  LBMK_REQUIRE(d.batch >= 1 && d.batch <= 65535, "igemm: batch %d", d.batch);
  LBMK_REQUIRE((d.KH == 1 && d.KW == 1) || (d.KH == 3 && d.KW == 3), "igemm: kernel %dx%d", d.KH, d.KW);
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
  LBMK_REQUIRE(d.act >= 0 && d.act <= 2, "igemm: act %d", d.act);
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

  int tile = d.tile;
  if (tile < 0) {
    if (d.act != LBMK_ACT_GEGLU && d.N <= 16) tile = 2;
    else {
      // Cost model over the production tiles: workgroups run in rounds of (CUs x resident blocks per
      // CU); a round costs resident * bm * bn / rate, padded rows / columns included.  `rate` is the
      // measured full-occupancy throughput of the tile shape (TFLOP/s, mid-size K): the 16-wave
      // double-buffered tiles move half the L2->LDS bytes per FLOP of the 128x128 tile.
      struct Cand { int tile, bm, bn, resident; float rate; };
      static const Cand cands[] = {{7, 256, 256, 1, 1150.f}, {8, 512, 128, 1, 1050.f}, {9, 256, 128, 1, 880.f},
                                   {0, 128, 128, 2, 800.f},  {1, 128, 64, 3, 700.f},   {3, 64, 64, 5, 450.f}};
      const int cus = lbmk_num_cus();
      double best = 1e300;
      tile = 0;
      for (const Cand& c : cands) {
        const long blocks = (long)((d.M + c.bm - 1) / c.bm) * ((d.N + c.bn - 1) / c.bn) * d.batch;
        const long slots = (long)cus * c.resident;
        const long rounds = (blocks + slots - 1) / slots;
        const double cost = (double)rounds * c.resident * c.bm * c.bn / c.rate;
        if (cost < best) { best = cost; tile = c.tile; }
      }
    }
  }
  if (tile == 100) { launch<256, 256, 4, 4, true, 1>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 102) { launch<256, 256, 4, 4, true, 3>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 110) { launch<512, 128, 8, 2, true, 1>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 111) { launch<512, 128, 8, 2, true, 2>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 112) { launch<512, 128, 8, 2, true, 3>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 115) { launch<512, 128, 8, 2, true, 5>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 116) { launch<256, 256, 4, 4, true, 5>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  if (tile == 101) { launch<256, 256, 4, 4, true, 2>(d, (hipStream_t)stream); return lbmk_check_launch("igemm"); }
  LBMK_REQUIRE(tile >= 0 && tile < kNumTiles, "igemm: tile %d", tile);
  if (d.act == LBMK_ACT_GEGLU) LBMK_REQUIRE(tile != 2, "igemm: GEGLU needs a tile with >= 32 columns per wave");
  hipStream_t s = (hipStream_t)stream;
  switch (tile) {
    case 0: launch<128, 128, 2, 2, false>(d, s); break;
    case 1: launch<128, 64, 2, 2, false>(d, s); break;
    case 2: launch<256, 16, 4, 1, false>(d, s); break;
    case 3: launch<64, 64, 2, 2, false>(d, s); break;
    case 4: launch<256, 64, 4, 1, false>(d, s); break;
    case 5: launch<256, 256, 2, 4, true>(d, s); break;
    case 6: launch<256, 128, 4, 2, true>(d, s); break;
    case 7: launch<256, 256, 4, 4, true>(d, s); break;
    case 8: launch<512, 128, 8, 2, true>(d, s); break;
    case 9: launch<256, 128, 4, 4, true>(d, s); break;
    case 10: launch_k32<256, 256, 4, 4, 4>(d, s); break;
    case 11: launch_k32<256, 256, 4, 4, 5>(d, s); break;
    case 12: launch_persist<256, 256, 4, 4>(d, s); break;
    case 13: launch_persist<512, 128, 8, 2>(d, s); break;
  }
  return lbmk_check_launch("igemm");
}
