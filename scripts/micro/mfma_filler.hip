// What does a vector instruction cost beside MFMAs?  Every wave runs a fixed instruction stream (inline asm, nothing for
// the compiler to move): four v_mfma_f32_32x32x16_bf16 on four independent accumulators, each followed by K filler
// vector instructions (v_fma_f32 or v_exp_f32 on registers the MFMAs do not touch).  One workgroup per CU, W waves per
// SIMD.  Prints nanoseconds per MFMA slot of a SIMD (launch time / (iterations * 4 * W)) for K = 0 ... 10: with a fixed
// clock, K fillers are free while that number stays at the K = 0 value.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_filler.hip -o scripts/micro/filler.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %12, %" #i "\n"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n"

template <int K, int VOP>
__device__ __forceinline__ void slot(f32x16& acc, const bf16x8& a, const bf16x8& b, float (&v)[8], float c) {
  // operands: %0 acc, %1 a, %2 b, %3 dummy, %4..%11 v[0..7], %12 c
#define BODY(F)                                                                                                       \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" F                                                          \
               : "+v"(acc), "+v"(*(bf16x8*)&a), "+v"(*(bf16x8*)&b), "+v"(c), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]),      \
                 "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])                                           \
               : "v"(c))
  if constexpr (VOP == 0) {
    if constexpr (K == 0) BODY("");
    if constexpr (K == 1) BODY(FMA(4));
    if constexpr (K == 2) BODY(FMA(4) FMA(5));
    if constexpr (K == 3) BODY(FMA(4) FMA(5) FMA(6));
    if constexpr (K == 4) BODY(FMA(4) FMA(5) FMA(6) FMA(7));
    if constexpr (K == 5) BODY(FMA(4) FMA(5) FMA(6) FMA(7) FMA(8));
    if constexpr (K == 6) BODY(FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9));
    if constexpr (K == 8) BODY(FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11));
    if constexpr (K == 10) BODY(FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11) FMA(4) FMA(5));
    if constexpr (K == 12) BODY(FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11) FMA(4) FMA(5) FMA(6) FMA(7));
  } else {
    if constexpr (K == 0) BODY("");
    if constexpr (K == 1) BODY(EXP(4));
    if constexpr (K == 2) BODY(EXP(4) EXP(5));
    if constexpr (K == 3) BODY(EXP(4) EXP(5) EXP(6));
    if constexpr (K == 4) BODY(EXP(4) EXP(5) EXP(6) EXP(7));
    if constexpr (K == 5) BODY(EXP(4) EXP(5) EXP(6) EXP(7) EXP(8));
    if constexpr (K == 6) BODY(EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9));
    if constexpr (K == 8) BODY(EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) EXP(11));
    if constexpr (K == 10) BODY(EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) EXP(11) EXP(4) EXP(5));
    if constexpr (K == 12) BODY(EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) EXP(11) EXP(4) EXP(5) EXP(6) EXP(7));
  }
#undef BODY
}

template <int K, int VOP>
__global__ __launch_bounds__(768) void body(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * ((threadIdx.x * 7 + e) % 97)); b[e] = (__bf16)(0.002f * ((threadIdx.x + 3 * e) % 89)); }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = -0.01f * (threadIdx.x & 63) - e;
  for (int it = 0; it < iters; ++it) {
    slot<K, VOP>(acc[0], a, b, v, 0.999f);
    slot<K, VOP>(acc[1], a, b, v, 0.999f);
    slot<K, VOP>(acc[2], a, b, v, 0.999f);
    slot<K, VOP>(acc[3], a, b, v, 0.999f);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int VOP>
float run(float* out, int W, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((body<K, VOP>), dim3(256), dim3(256 * W), 0, 0, out, 500);
  hipEventRecord(e0);
  hipLaunchKernelGGL((body<K, VOP>), dim3(256), dim3(256 * W), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / ((float)iters * 4 * W);
}

template <int VOP>
void table(float* out) {
  printf("%s fillers per MFMA:        0      1      2      3      4      5      6      8     10     12\n", VOP ? "v_exp_f32" : "v_fma_f32");
  for (int W : {1, 2, 3}) {
    const int it = 40000 / W;
    printf("  %d wave(s)/SIMD, ns/slot: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f\n", W,
           run<0, VOP>(out, W, it), run<1, VOP>(out, W, it), run<2, VOP>(out, W, it), run<3, VOP>(out, W, it), run<4, VOP>(out, W, it),
           run<5, VOP>(out, W, it), run<6, VOP>(out, W, it), run<8, VOP>(out, W, it), run<10, VOP>(out, W, it), run<12, VOP>(out, W, it));
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  table<0>(out);
  table<1>(out);
  return 0;
}
