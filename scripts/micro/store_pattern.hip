// Store-pattern micro-benchmark: every wave writes a 64-row x 128-column bf16 tile of a row-major [M][N] matrix
// with 16-byte stores, 16 per lane, in four lane -> address maps.  Prints microseconds per pass and GB/s.
//   seg = bytes that the lanes of one store instruction write contiguously in one matrix row (64 ... 256)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int SEG>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, int M, int N, int tiles_n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * 256 + wm * 64, n0 = tile_n * 256 + wn * 128;
  constexpr int LPR = SEG / 16;        // lanes per row segment
  constexpr int RPI = 64 / LPR;        // rows per instruction
  constexpr int SPR = 256 / SEG;       // segments per 128-column row
  const u32x4 v = {(unsigned)lane, (unsigned)wave, (unsigned)blockIdx.x, 0x3f803f80u};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // instruction k covers RPI rows x SEG bytes: walk the 64 x 256-byte tile segment-column-major inside row blocks
    const int blk = k / SPR, seg = k % SPR;          // row block of RPI rows (64 / RPI blocks ... ) x segment
    const int row = (blk * RPI) % 64 + lane / LPR;
    const int colb = seg * SEG + (lane % LPR) * 16;  // byte column inside the wave's 256-byte row
    *(u32x4*)((char*)(out + (size_t)(m0 + row) * N + n0) + colb) = v;
  }
}

int main() {
  const int M = 16 * 256 * 256, N = 512;   // the 512-channel 256^2 x 16 activation
  unsigned short* out;
  hipMalloc(&out, (size_t)M * N * 2);
  const int tiles_n = N / 256, blocks = (M / 256) * tiles_n;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, M, N, tiles_n);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, M, N, tiles_n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-28s %8.1f us  %7.2f TB/s  (%.2f us per 256-block round)\n", name, ms * 1e3, (double)M * N * 2 / ms / 1e9, ms * 1e3 / (blocks / 256.0));
  };
  run(store_kernel<64>, "16 rows x 64 B per instr");
  run(store_kernel<128>, "8 rows x 128 B per instr");
  run(store_kernel<256>, "4 rows x 256 B per instr");
  // Is the floor per CU or chip-wide?  The same 128 KB per workgroup with only `g` workgroups (one per CU) launched.
  for (int g : {256, 128, 64, 32, 8}) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_kernel<64>, dim3(g), dim3(512), 0, 0, out, M, N, tiles_n);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<64>, dim3(g), dim3(512), 0, 0, out + (size_t)i * g * 65536, M, N, tiles_n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%3d workgroups x 128 KB: %6.2f us per launch (launch overhead included), %5.2f TB/s\n", g, ms * 1e3, (double)g * 131072 / ms / 1e9);
  }
  return 0;
}
