// Do matrix and vector instructions of DIFFERENT waves on one SIMD overlap, and what does an interleaved stream of ONE
// wave cost?  One workgroup per CU (256 workgroups), W waves per SIMD (blockDim = 256 * W), every wave runs ITER
// iterations of a body chosen by `mode` and the wave's role:
//   mode 0: every wave: NM MFMAs back to back, then NV vector ops back to back         (the attention loop's shape)
//   mode 1: every wave: MFMAs only                    mode 2: every wave: vector ops only
//   mode 3: waves 0-3 of the workgroup MFMAs only, the others vector ops only         (roles split between SIMD partners)
//   mode 4: every wave: one MFMA followed by NV/NM vector ops, NM times               (interleaved stream)
// Prints nanoseconds per iteration (HIP events over the launch / ITER) and s_memtime ticks of wave 0 per iteration; a SIMD
// runs W waves' iterations in that time.  VOP selects exp2 (1) or fma (0).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NV, int VOP, int NACC, int mode>
__global__ __launch_bounds__(768) void body(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = 0.01f * (threadIdx.x & 63) + e;
  const bool do_m = mode == 0 || mode == 1 || mode == 4 || (mode == 3 && wave < 4);
  const bool do_v = mode == 0 || mode == 2 || mode == 4 || (mode == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (mode == 4) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV / NM; ++k) {
          const int j = (m * (NV / NM) + k) & 7;
          if (VOP) v[j] = __builtin_amdgcn_exp2f(v[j]); else v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
        }
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          if (VOP) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]); else v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        }
      }
    }
    asm volatile("" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int VOP, int NACC>
void sweep() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("NM = %d MFMA (32x32x16 bf16) over %d accumulators, NV = %d x %s per iteration\n", NM, NACC, NV, VOP ? "v_exp_f32" : "v_fma_f32");
  for (int W : {1, 2, 3}) {
    printf("  %d wave(s)/SIMD, ns per iteration [s_memtime ticks]:", W);
    auto one = [&](auto kern, int mode) {
      if (mode == 3 && W != 2) { printf("               -"); return; }
      hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 0, 0, out, cyc, 200);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 0, 0, out, cyc, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("  m%d %5.0f [%5.0f]", mode, ms * 1e6 / iters, (double)c / iters);
    };
    one(body<NM, NV, VOP, NACC, 0>, 0); one(body<NM, NV, VOP, NACC, 1>, 1); one(body<NM, NV, VOP, NACC, 2>, 2);
    one(body<NM, NV, VOP, NACC, 3>, 3); one(body<NM, NV, VOP, NACC, 4>, 4);
    printf("\n");
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("modes: m0 MFMAs then vector ops | m1 MFMA only | m2 vector only | m3 waves 0-3 MFMA, 4-7 vector | m4 interleaved\n");
  sweep<16, 96, 0, 2>();
  sweep<16, 96, 0, 4>();
  sweep<16, 48, 0, 4>();
  sweep<16, 96, 1, 2>();
  sweep<16, 96, 1, 4>();
  sweep<16, 32, 1, 4>();
  sweep<16, 160, 0, 4>();
  return 0;
}
